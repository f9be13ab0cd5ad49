#!/bin/bash
# Round 4, GPU call 2: the recomputed-M Farneback iteration kernel — parity, then A/B against the M-in-HBM kernel on one box.
mkdir -p gpurun_out/r4_farn1
cd /root/repo
free -g | head -2 > gpurun_out/r4_farn1/box.txt
timeout 900 python -m pytest tests/test_farneback_gpu.py -x -q > gpurun_out/r4_farn1/pytest_farn.log 2>&1
tail -5 gpurun_out/r4_farn1/pytest_farn.log
for v in 0 16 0 16; do
  timeout 300 python bench.py --algo farn --steps 3 --warmup 1 --variant $v --no-cpu-baseline --no-others --no-pcie > gpurun_out/r4_farn1/bench_farn_variant${v}_$RANDOM.json 2>> gpurun_out/r4_farn1/err.log
done
grep -h -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/r4_farn1/bench_farn_variant*.json
