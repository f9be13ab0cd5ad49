#!/bin/bash
# round 6, GPU call 18: head kernel's bicubic weights without the chain's last select (|x| clamped to 2) + one unsigned window test
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_18; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_device_math_gpu.py tests/test_tvl1_gpu.py tests/test_content_classes_gpu.py -q -m gpu -x -k "bicubic or tvl1 or content" 2>&1 | tail -4
for rep in 1 2 3; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/bench_$rep.json 2> $O/bench_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$rep.json").read().strip().splitlines()[-1])
print("rep $rep:", round(d["value"],1), "pairs/s  parity", d.get("parity_check",{}).get("max_abs"))
PY
done
