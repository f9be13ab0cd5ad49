#!/bin/bash
# round 2, GPU call 14: start stagger of the step kernel's first workgroups (0 / 4k / 8k / 16k / 32k cycles)
mkdir -p gpurun_out/r2n; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n
cd $R
( SWEEP="0:4:0:0:3:0:0:0:0,0:4:0:0:3:0:0:0:4000,0:4:0:0:3:0:0:0:8000,0:4:0:0:3:0:0:0:16000,0:4:0:0:3:0:0:0:32000,0:4:0:0:3:0:0:0:0,0:4:0:0:3:0:0:0:12000" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_stagger.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_stagger.log | cut -c1-420
