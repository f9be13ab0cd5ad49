#!/bin/bash
# round 2, GPU call 16: XCD-aware workgroup -> tile mapping in the Farneback / Brox kernels and the TVL1 warp kernel
# (default build) against a build without it (denseflow_amd/lib/variants/libdfx_noxcd.so, -DDFX_XCD_REMAP=0)
mkdir -p gpurun_out/r2p; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p
cd $R
( timeout 600 python -m pytest tests/test_farneback_gpu.py tests/test_brox_gpu.py tests/test_tvl1_gpu.py -m gpu -q -k "not config5 and not tile_geometry and not fused_kernel_equals" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
V=$R/denseflow_amd/lib/variants/libdfx_noxcd.so
for L in xcd noxcd xcd noxcd; do
  if [ $L = noxcd ]; then export DFX_LIBRARY=$V; else unset DFX_LIBRARY; fi
  ( ALGO=farn SWEEP="0:0:0:0,0:0:0:0" timeout 200 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/farn_$L.log 2>&1; echo "farn $L"; grep -v amdgpu.ids $O/farn_$L.log | cut -c1-190
  ( ALGO=brox SWEEP="0:0:0:0,0:0:0:0" timeout 200 python scripts/sweep_tvl1.py 1920 1080 33 ) > $O/brox_$L.log 2>&1; echo "brox $L"; grep -v amdgpu.ids $O/brox_$L.log | cut -c1-190
  ( SWEEP="0:4:0:0,0:4:0:0" SWEEP_LEVELS=1 timeout 200 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/tvl1_$L.log 2>&1; echo "tvl1 $L"; grep -v amdgpu.ids $O/tvl1_$L.log | cut -c1-420
done
unset DFX_LIBRARY
( ALGO=brox SWEEP="0:0:0:0" timeout 200 python scripts/sweep_tvl1.py 3840 2160 9 ) > $O/brox4k_xcd.log 2>&1; grep -v amdgpu.ids $O/brox4k_xcd.log | cut -c1-190
( DFX_LIBRARY=$V ALGO=brox SWEEP="0:0:0:0" timeout 200 python scripts/sweep_tvl1.py 3840 2160 9 ) > $O/brox4k_noxcd.log 2>&1; grep -v amdgpu.ids $O/brox4k_noxcd.log | cut -c1-190
