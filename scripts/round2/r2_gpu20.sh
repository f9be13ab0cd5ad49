#!/bin/bash
# round 2, GPU call 20: can two handles of ONE process overlap like two processes do (+6.5 %)?  Streams of different priority
# live on different hardware queues.
mkdir -p gpurun_out/r2t; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t
cd $R
python -c "import torch; print(torch.cuda.get_device_properties(0)); import ctypes; L=ctypes.CDLL('libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); print('prio range rc', L.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), lo.value, hi.value)" 2>&1 | tail -2
for P in "0,-1" "1,-1" "0,0" "-1,-1"; do
( PRIOS="$P" LANES="2,2" timeout 200 python scripts/multi_lane_probe.py 1920 1080 240 ) > $O/lanes_prio.log 2>&1; echo "PRIOS=$P"; grep -v amdgpu.ids $O/lanes_prio.log | cut -c1-160
done
( GPU_MAX_HW_QUEUES=8 PRIOS="0,-1,1" LANES="2,3" timeout 200 python scripts/multi_lane_probe.py 1920 1080 240 ) > $O/lanes_prio8.log 2>&1; echo "HWQ=8 PRIOS=0,-1,1"; grep -v amdgpu.ids $O/lanes_prio8.log | cut -c1-160
