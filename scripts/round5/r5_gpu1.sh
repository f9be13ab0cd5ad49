#!/bin/bash
# round 5, GPU call 1: the float hypot readings — element-wise exactness, flow-level parity, and the rate of each mode
set -u
mkdir -p gpurun_out/r5_1
python -m pytest tests/test_device_math_gpu.py tests/test_tvl1_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5_1/pytest.log
cat gpurun_out/r5_1/pytest.log
for m in exact sqrt libm; do
  python bench.py --math $m --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > gpurun_out/r5_1/bench_$m.json 2> gpurun_out/r5_1/bench_$m.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_1/bench_$m.json").read().strip().splitlines()[-1])
print("$m", round(d["value"],1), "pairs/s  iters/pair", d["config"]["mean_inner_iterations_per_pair"], "frac", round(d["roofline"]["frac"],3), "launch us", round(d["roofline"]["avg_launch_us"],1))
PY
done
