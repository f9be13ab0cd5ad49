#!/bin/bash
# round 5, GPU call 11: device batch size on the headline clip (299 pairs): 129 + 129 + 41 (auto) vs 150 + 149 vs 299
O=gpurun_out/r5_11; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
for r in 1 2; do for mb in 0 150 299; do
  python bench.py --max-batch $mb --steps 5 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/b_$mb.json 2> $O/b_$mb.err
  python - <<PY
import json
a=json.loads(open("gpurun_out/r5_11/b_$mb.json").read().strip().splitlines()[-1])
print("max_batch $mb", round(a["value"],1), "launch us", round(a["roofline"]["avg_launch_us"],1), "pairs/launch", a["config"]["pairs_per_launch"])
PY
done; done
python bench.py --algo farn --max-batch 299 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('farn 299', d['value'])"
python bench.py --algo farn --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('farn auto', d['value'])"
