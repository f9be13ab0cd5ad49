#!/bin/bash
# bench lines for both algorithms + rocprofv3 kernel-trace stats of the same commands
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
( timeout 900 python bench.py ) > gpurun_out/bench_tvl1.log 2>&1; echo "bench tvl1 rc=$?"; tail -1 gpurun_out/bench_tvl1.log
( timeout 900 python bench.py --algo farn ) > gpurun_out/bench_farn.log 2>&1; echo "bench farn rc=$?"; tail -1 gpurun_out/bench_farn.log
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_tvl1 -o tvl1 -- python $R/bench.py --steps 1 --warmup 1 --frames 100 --no-cpu-baseline ) > $R/gpurun_out/rocprof_bench_tvl1.log 2>&1; echo "rocprof tvl1 rc=$?"
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_farn -o farn -- python $R/bench.py --algo farn --steps 1 --warmup 1 --frames 100 --no-cpu-baseline ) > $R/gpurun_out/rocprof_bench_farn.log 2>&1; echo "rocprof farn rc=$?"
cd $R
for a in tvl1 farn; do F=gpurun_out/prof_bench_$a/${a}_kernel_stats.csv; [ -f $F ] && grep -E "k_tvl1|k_farn|k_u8|k_pyr|k_cent|Name" $F | cut -c1-160; rm -f gpurun_out/prof_bench_$a/*kernel_trace.csv; done
tail -2 gpurun_out/rocprof_bench_tvl1.log | cut -c1-300
