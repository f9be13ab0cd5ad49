#!/bin/bash
# round 2, GPU call 5: full GPU suite, bench.py (new legs), PMC calibration, end-to-end CLI rates with submit/wait
mkdir -p gpurun_out/r2e; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
( timeout 900 python bench.py ) > $O/bench_tvl1.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_tvl1.log | cut -c1-3000
cd /tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
 ( timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $O/cal_$CNT -o p -- python $R/scripts/pmc_calibrate.py 2 ) > $O/cal_$CNT.log 2>&1; echo "cal $CNT rc=$?"
 python $R/scripts/sq_summary.py $O/cal_$CNT k_calib > $O/cal_$CNT.json 2>>$O/cal_$CNT.log; rm -rf $O/cal_$CNT; cat $O/cal_$CNT.json
done
cd $R
( ALGOS=tvl1,farn timeout 600 python scripts/e2e_cli_rate.py 1920 1080 513 ) > $O/e2e_1080p.log 2>&1; echo "e2e 1080p rc=$?"; grep -v amdgpu.ids $O/e2e_1080p.log
( ALGOS=tvl1 timeout 600 python scripts/e2e_cli_rate.py 224 224 300 32 ) > $O/e2e_224.log 2>&1; echo "e2e 224 rc=$?"; grep -v amdgpu.ids $O/e2e_224.log
