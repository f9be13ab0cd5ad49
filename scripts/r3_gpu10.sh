#!/bin/bash
# round 3, GPU session 10: the save stage writes device-encoded files in parallel
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_shell.py -m gpu -x -q > $O/pytest_shell.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_shell.log
for rep in 1 2; do CONFIGS="device" ALGOS=tvl1,farn timeout 600 python scripts/e2e_cli_rate.py 224 224 300 64 2>&1 | grep -v amdgpu.ids; done | tee $O/e2e_224x64clips.log
for rep in 1 2; do DF_ENCODE_THREADS=1 CONFIGS="device" ALGOS=tvl1 timeout 600 python scripts/e2e_cli_rate.py 224 224 300 64 2>&1 | grep -v amdgpu.ids; done | tee $O/e2e_224x64clips_one_writer.log
CONFIGS="device" ALGOS=tvl1,farn timeout 900 python scripts/e2e_cli_rate.py 1920 1080 3073 2>&1 | grep -v amdgpu.ids | tee $O/e2e_1080p_3073.log
CONFIGS="dev-bound/host-jpeg" ALGOS=tvl1 timeout 900 python scripts/e2e_cli_rate.py 1920 1080 3073 2>&1 | grep -v amdgpu.ids | tee -a $O/e2e_1080p_3073.log
