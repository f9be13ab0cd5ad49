#!/bin/bash
# round 2, GPU call 15: warp kernel variants — 8 strips per workgroup (cheap no-op launches), gradient of I1 at the taps
mkdir -p gpurun_out/r2o; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2o
cd $R
( timeout 300 python -m pytest tests/test_tvl1_gpu.py -m gpu -q -k "warp_kernel_variants or bit_exact or single_pair" ) > $O/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -3 $O/pytest_a.log
( DFX_TVL1_WARP_OTF=1 timeout 400 python -m pytest tests/test_tvl1_gpu.py tests/test_edge_sizes_gpu.py -m gpu -q -k "not tile_geometry and not warp_kernel_variants and not farneback and not brox" ) > $O/pytest_otf.log 2>&1; echo "pytest otf rc=$?"; tail -3 $O/pytest_otf.log
# fields impl:K:B:TH:geom:sor:skip0:polyrows:spw:otf
( SWEEP="0:4:0:0:3:0:1:16:1:0,0:4:0:0:3:0:1:16:8:0,0:4:0:0:3:0:1:16:8:1,0:4:0:0:3:0:1:16:8:2,0:4:0:0:3:0:1:16:1:0,0:4:0:0:3:0:1:16:8:0,0:4:0:0:3:0:1:16:8:1" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_warp.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_warp.log | cut -c1-430
( ITERS=0 SWEEP="0:4:0:0:3:0:1:16:8:0,0:4:0:0:3:0:1:16:8:1,0:4:0:0:3:0:1:16:8:2,0:4:0:0:3:0:1:16:8:0" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_warponly.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_warponly.log | cut -c1-430
( SWEEP="0:4:0:0:3:0:1:16:1:0,0:4:0:0:3:0:1:16:8:0,0:4:0:0:3:0:1:16:8:1,0:4:0:0:3:0:1:16:1:0" timeout 300 python scripts/sweep_tvl1.py 224 224 300 ) > $O/sweep_warp_224.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_warp_224.log | cut -c1-230
