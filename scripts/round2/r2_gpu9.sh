#!/bin/bash
# round 2, GPU call 9: dedicated warp kernel — parity, A/B, and the share of the warps in a pair's time
mkdir -p gpurun_out/r2i; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i
cd $R
( timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_edge_sizes_gpu.py tests/test_async_gpu.py tests/test_quant_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for S in 0 1 0 1; do
( DFX_TVL1_SPLIT_WARP=$S SWEEP="0:4:0:0" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_split$S.log 2>&1; echo "split=$S"; grep -v amdgpu.ids $O/sweep_split$S.log | cut -c1-330
done
for S in 0 1; do
( DFX_TVL1_SPLIT_WARP=$S ITERS=0 SWEEP="0:4:0:0" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_warponly_split$S.log 2>&1; echo "warps only split=$S"; grep -v amdgpu.ids $O/sweep_warponly_split$S.log | cut -c1-330
( DFX_TVL1_SPLIT_WARP=$S SWEEP="0:4:0:0" timeout 300 python scripts/sweep_tvl1.py 224 224 300 ) > $O/sweep224_split$S.log 2>&1; grep -v amdgpu.ids $O/sweep224_split$S.log | cut -c1-200
done
