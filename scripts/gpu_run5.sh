#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log
( ALGO=farn SWEEP_LEVELS=1 SWEEP="0:0:16,0:0:4" timeout 600 python scripts/sweep_tvl1.py ) > gpurun_out/sweep_farn.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids gpurun_out/sweep_farn.log
