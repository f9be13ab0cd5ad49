#!/bin/bash
# round 6, GPU call 12: where a 224x224 pair's time goes per pyramid level, and what fuse_k does there
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_12; mkdir -p $O; cd $R
for k in 4 3 5 6; do
  DFX_BENCH_LEVELS=1 python bench.py --width 224 --height 224 --clips 64 --fuse-k $k --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/b224_k$k.json 2> $O/b224_k$k.err
  python - <<PY
import json
d=json.loads(open("$O/b224_k$k.json").read().strip().splitlines()[-1])
print("K=$k:", round(d["value"],1), "pairs/s  frac", round(d["roofline"]["frac"],3), "useful", d["roofline"].get("useful_frac"), "noop", d["config"].get("noop_step_fraction"))
PY
  grep "levels:" $O/b224_k$k.err
done
