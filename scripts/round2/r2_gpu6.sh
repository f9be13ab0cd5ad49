#!/bin/bash
# round 2, GPU call 6: CPU thread scan on the GPU box's host, PMC calibration, shell A/B sync vs submit/wait, remaining tests
mkdir -p gpurun_out/r2f; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f
cd $R
( timeout 600 python scripts/cpu_thread_scan.py 4 8 16 32 64 128 ) > $O/cpu_thread_scan.log 2>&1; echo "scan rc=$?"; cat $O/cpu_thread_scan.log | cut -c1-300
cd /tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
 ( timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $O/cal_$CNT -o p -- python $R/scripts/pmc_calibrate.py 2 ) > $O/cal_$CNT.log 2>&1; echo "cal $CNT rc=$?"
 python $R/scripts/sq_summary.py $O/cal_$CNT k_calib > $O/cal_$CNT.json 2>>$O/cal_$CNT.log; rm -rf $O/cal_$CNT; cat $O/cal_$CNT.json
done
cd $R
for S in 1 0; do
 if [ $S = 1 ]; then export DF_SYNC_FLOW=1; else unset DF_SYNC_FLOW; fi
 ( ALGOS=tvl1,farn timeout 600 python scripts/e2e_cli_rate.py 1920 1080 513 ) > $O/e2e_1080p_sync$S.log 2>&1; echo "e2e 1080p sync=$S rc=$?"; grep "device" $O/e2e_1080p_sync$S.log
 ( ALGOS=tvl1 timeout 600 python scripts/e2e_cli_rate.py 224 224 300 32 ) > $O/e2e_224_sync$S.log 2>&1; echo "e2e 224 sync=$S rc=$?"; grep "device" $O/e2e_224_sync$S.log
done
unset DF_SYNC_FLOW
( timeout 1200 python -m pytest tests/test_host_shell.py tests/test_opencv_pin.py tests/test_prepare_gpu.py tests/test_quant_gpu.py tests/test_tvl1_gpu.py -m gpu -q ) > $O/pytest_rest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_rest.log
