#!/bin/bash
# rocprofv3 kernel trace (+stats) of a short sweep run; CSV output copied to gpurun_out/prof_<tag>
TAG=${1:-r1}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
( SWEEP="${SWEEP:-0:4:8}" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o tvl1 -- python $R/scripts/sweep_tvl1.py ${SIZE:-1920 1080 17} ) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; find gpurun_out/prof_$TAG -type f | head; tail -3 gpurun_out/rocprof_$TAG.log
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cat $F | cut -c1-200
