#!/bin/bash
# round 2, GPU call 8: trapezoid row layout of the TVL1 step kernel — parity and sweep
mkdir -p gpurun_out/r2h; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h
cd $R
( timeout 600 python -m pytest tests/test_tvl1_gpu.py -m gpu -x -q -k "fused_kernel" ) > $O/pytest_tvl1.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_tvl1.log
( SWEEP="0:4:0:0,0:4:0:321,0:5:0:321,0:6:0:321,0:3:0:321,0:4:0:0,0:4:0:321" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_1080p.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids $O/sweep_1080p.log | cut -c1-330
( EPS=1e-9 SWEEP="0:4:0:321" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_noconv.log 2>&1; grep -v amdgpu.ids $O/sweep_noconv.log | cut -c1-330
( SWEEP="0:4:0:0,0:4:0:321,0:3:0:321" timeout 300 python scripts/sweep_tvl1.py 224 224 300 ) > $O/sweep_224.log 2>&1; grep -v amdgpu.ids $O/sweep_224.log | cut -c1-200
