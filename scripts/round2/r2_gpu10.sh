#!/bin/bash
# round 2, GPU call 10: warp kernel register budgets; warps-only timing
mkdir -p gpurun_out/r2j; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j
cd $R
( timeout 600 python -m pytest tests/test_tvl1_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for W in 5 7 8 5 7; do
( DFX_TVL1_WARP_WPS=$W SWEEP="0:4:0:0" timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_wps$W.log 2>&1; echo "wps=$W"; grep -v amdgpu.ids $O/sweep_wps$W.log | cut -c1-200
done
for W in 5 7 8; do
( DFX_TVL1_WARP_WPS=$W ITERS=0 SWEEP="0:4:0:0" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_warponly_wps$W.log 2>&1; echo "warps only wps=$W"; grep -v amdgpu.ids $O/sweep_warponly_wps$W.log | cut -c1-330
done
( DFX_TVL1_WARP_WPS=7 timeout 300 python -m pytest tests/test_tvl1_gpu.py -m gpu -x -q -k "bit_exact or fused_kernel" ) > $O/pytest_wps7.log 2>&1; echo "pytest wps7 rc=$?"; tail -2 $O/pytest_wps7.log
