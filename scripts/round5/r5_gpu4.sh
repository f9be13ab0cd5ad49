#!/bin/bash
# round 5, GPU call 4: threshold-step trim (A/B by bench), the three readings against each other at 1080p, shell trace
O=gpurun_out/r5_4; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
make -s host > $O/make_host.log 2>&1
timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_edge_sizes_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for r in 1 2; do for m in exact libm; do
  python bench.py --math $m --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_$m.json 2> $O/bench_$m.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_4/bench_$m.json").read().strip().splitlines()[-1])
print("$m", round(d["value"],1), "pairs/s  launch us", round(d["roofline"]["avg_launch_us"],1))
PY
done; done
python scripts/round5/hypot_readings_eval.py 300 > $O/readings_1080p.md 2> $O/readings.err; cat $O/readings_1080p.md; tail -2 $O/readings.err
python scripts/e2e_trace.py 1920 1080 1537 farn > $O/e2e_trace_farn.log 2>&1; grep -c . $O/e2e_trace_farn.log; grep "calc:\|submitted\|engine" $O/e2e_trace_farn.log | head -40
