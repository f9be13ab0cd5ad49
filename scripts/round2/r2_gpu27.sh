#!/bin/bash
# round 2, GPU call 27: the Brox fused-SOR tile / sweep configurations again, now that the XCD-aware mapping serves the halo
# re-reads from L2 (round 1: 64x64x5 164, 64x64x3 135, 64x32x2 138, 128x32x2 133 pairs/s)
mkdir -p gpurun_out/r2z; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2z
cd $R
( ALGO=brox SWEEP="0:0:0:0,0:0:0:643,0:0:0:642,0:0:0:64,0:0:0:128,0:0:0:0" timeout 300 python scripts/sweep_tvl1.py 1920 1080 33 ) > $O/sweep_brox_cfg.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_brox_cfg.log | cut -c1-200
