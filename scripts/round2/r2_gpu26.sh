#!/bin/bash
# round 2, GPU call 26: 64x96 tile on 12 waves (one workgroup per CU, 80 % of the tile owned) against the 64x32 default
mkdir -p gpurun_out/r2y; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2y
cd $R
( timeout 400 python -m pytest tests/test_tvl1_gpu.py -m gpu -q -x -k "fused_kernel_equals_simple" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
( SWEEP="0:4:0:0:3,0:4:0:961:3,0:4:0:0:3,0:4:0:961:3,0:6:0:961:3" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_96.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_96.log | cut -c1-420
