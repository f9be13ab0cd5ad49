#!/bin/bash
# round 3, GPU session 20: what the 2048-pair automatic batch costs at engine start-up (224x224), and the shell's joined flow
# stage on a longer list
O=gpurun_out/r3t; mkdir -p $O
for a in tvl1 farn brox; do python scripts/engine_startup.py 224 224 $a 512 2048 0 2>&1 | grep -v amdgpu.ids; done | tee $O/engine_startup_224.txt
python scripts/engine_startup.py 1920 1080 tvl1 0 2>&1 | grep -v amdgpu.ids | tee -a $O/engine_startup_224.txt
timeout 300 python -m pytest tests/test_segments_gpu.py -x -q 2>&1 | tail -2
ALGOS=tvl1 CONFIGS="device" timeout 600 python scripts/e2e_cli_rate.py 224 224 300 192 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x192clips_joined.log
ALGOS=tvl1 DF_NO_JOIN=1 CONFIGS="device" timeout 600 python scripts/e2e_cli_rate.py 224 224 300 192 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x192clips_not_joined.log
