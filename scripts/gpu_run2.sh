#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
( timeout 900 python scripts/sweep_tvl1.py ) > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"; cat gpurun_out/sweep.log | grep -v amdgpu.ids
