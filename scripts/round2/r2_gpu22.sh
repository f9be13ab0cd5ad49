#!/bin/bash
# round 2, GPU call 22: fused warp mode (the step kernel takes the warp into registers and runs the first step of its loop)
mkdir -p gpurun_out/r2v; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v
cd $R
( DFX_TVL1_FUSE_WARP=1 timeout 500 python -m pytest tests/test_tvl1_gpu.py tests/test_edge_sizes_gpu.py -m gpu -q -x -k "not farneback and not brox and not tile_geometry" ) > $O/pytest_fuse.log 2>&1; echo "pytest fuse rc=$?"; tail -5 $O/pytest_fuse.log | cut -c1-300
( SWEEP="0:4:0:0:3:0:1:16:0,0:4:0:0:3:0:1:16:1,0:4:0:0:3:0:1:16:0,0:4:0:0:3:0:1:16:1" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_fuse.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_fuse.log | cut -c1-420
( SWEEP="0:4:0:0:3:0:1:16:0,0:4:0:0:3:0:1:16:1,0:4:0:0:3:0:1:16:0,0:4:0:0:3:0:1:16:1" timeout 300 python scripts/sweep_tvl1.py 224 224 300 ) > $O/sweep_fuse_224.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_fuse_224.log | cut -c1-220
