#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( SWEEP_LEVELS=1 SWEEP="0:4:8,0:4:1,0:2:8" timeout 600 python scripts/sweep_tvl1.py ) > gpurun_out/sweep_levels.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids gpurun_out/sweep_levels.log
cd /tmp && ( SWEEP="0:4:8" timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1a -o tvl1 -- python $GRAFT_REPO_ROOT/scripts/sweep_tvl1.py 1920 1080 17 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r1a | head -20; find gpurun_out/prof_r1a -name "*kernel_stats*" | head -1 | xargs -r head -20
