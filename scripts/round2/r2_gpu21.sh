#!/bin/bash
# round 2, GPU call 21: is it the blocking hipEventSynchronize of one host thread that keeps the other handle's launches out?
mkdir -p gpurun_out/r2u; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u
cd $R
for PW in 0 20 100; do
( DFX_POLL_WAIT=$PW LANES="1,2,3,2" timeout 200 python scripts/multi_lane_probe.py 1920 1080 240 ) > $O/lanes_poll$PW.log 2>&1; echo "DFX_POLL_WAIT=$PW"; grep -v amdgpu.ids $O/lanes_poll$PW.log | cut -c1-160
done
( GPU_MAX_HW_QUEUES=8 DFX_POLL_WAIT=20 LANES="2,3" timeout 200 python scripts/multi_lane_probe.py 1920 1080 240 ) > $O/lanes_poll20_hwq8.log 2>&1; echo "HWQ=8 DFX_POLL_WAIT=20"; grep -v amdgpu.ids $O/lanes_poll20_hwq8.log | cut -c1-160
