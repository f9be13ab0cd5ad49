#!/bin/bash
# round 6, GPU call 1: the warp-and-head kernel — parity (TVL1 suites + content classes), then head vs two launches, alternating
set -u
O=gpurun_out/r6_1; mkdir -p $O
timeout 1500 python -m pytest tests/test_tvl1_gpu.py tests/test_content_classes_gpu.py tests/test_edge_sizes_gpu.py tests/test_bench_shaped_batch_gpu.py -q -m gpu -x -k "tvl1 or content" 2>&1 | tail -15 > $O/pytest.log
cat $O/pytest.log
for rep in 1 2; do for v in 0 64; do
  python bench.py --variant $v --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/bench_v${v}_$rep.json 2> $O/bench_v${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/bench_v${v}_$rep.json").read().strip().splitlines()[-1])
print("variant $v rep $rep:", round(d["value"],1), "pairs/s  iters/pair", d["config"]["mean_inner_iterations_per_pair"], "frac", round(d["roofline"]["frac"],3), "parity", d.get("parity_check",{}).get("max_abs"))
PY
done; done
