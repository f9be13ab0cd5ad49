#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
( SWEEP_LEVELS=1 SWEEP="${SWEEP:-0:4:8,0:6:8,0:3:8,0:4:16}" timeout 600 python scripts/sweep_tvl1.py ) > gpurun_out/sweep_levels.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids gpurun_out/sweep_levels.log
