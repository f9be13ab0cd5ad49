#!/bin/bash
# first GPU contact: smoke, gpu tests, short bench, rocprof summary
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" 
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/smoke.log; tail -25 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 1 --warmup 1 --frames 60 ) > gpurun_out/bench_60.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench_60.log
