#!/bin/bash
# round 2, GPU call 13: do two / three handles on one GPU (separate streams) overlap their HBM-bound and VALU-bound launches?
mkdir -p gpurun_out/r2m; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m
cd $R
( LANES="1,2,3,4,1,2" timeout 300 python scripts/multi_lane_probe.py 1920 1080 300 ) > $O/lanes_tvl1_1080p.log 2>&1; echo "tvl1 rc=$?"; grep -v amdgpu.ids $O/lanes_tvl1_1080p.log | cut -c1-200
( ALGO=farn LANES="1,2,3,1" timeout 300 python scripts/multi_lane_probe.py 1920 1080 300 ) > $O/lanes_farn_1080p.log 2>&1; echo "farn rc=$?"; grep -v amdgpu.ids $O/lanes_farn_1080p.log | cut -c1-200
( ALGO=brox LANES="1,2,1" REPS=1 timeout 300 python scripts/multi_lane_probe.py 1920 1080 66 ) > $O/lanes_brox_1080p.log 2>&1; echo "brox rc=$?"; grep -v amdgpu.ids $O/lanes_brox_1080p.log | cut -c1-200
( LANES="1,2,1,2" MAXB=512 timeout 300 python scripts/multi_lane_probe.py 224 224 600 ) > $O/lanes_tvl1_224.log 2>&1; echo "224 rc=$?"; grep -v amdgpu.ids $O/lanes_tvl1_224.log | cut -c1-200
