#!/bin/bash
# round 3, GPU session 19: the host shell joins queued short clips into one library call — CLI parity (joined vs DF_NO_JOIN=1) and
# the end-to-end rate on a list of 224x224 clips
O=gpurun_out/r3s; mkdir -p $O
timeout 1200 python -m pytest tests/test_host_shell.py tests/test_segments_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
CONFIGS="device" timeout 600 python scripts/e2e_cli_rate.py 224 224 300 64 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x64clips_joined.log
DF_NO_JOIN=1 CONFIGS="device" timeout 600 python scripts/e2e_cli_rate.py 224 224 300 64 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x64clips_not_joined.log
