#!/bin/bash
# round 3, GPU session 8: the shell with early engine creation: tests, start-up trace, long-clip end-to-end rates
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_shell.py -m gpu -x -q > $O/pytest_shell.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_shell.log
CONFIGS="device" ALGOS=tvl1 timeout 900 python scripts/e2e_cli_rate.py 1920 1080 1537 2>&1 | grep -v amdgpu.ids | tee $O/e2e_1080p_1537.log
CONFIGS="device" ALGOS=tvl1,farn timeout 900 python scripts/e2e_cli_rate.py 1920 1080 3073 2>&1 | grep -v amdgpu.ids | tee $O/e2e_1080p_3073.log
