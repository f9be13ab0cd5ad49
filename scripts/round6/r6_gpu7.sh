#!/bin/bash
# round 6, GPU call 7: TVL1 step launches per host poll — automatic with the new minimum of 2 against fixed 6 (rounds 1-5 at 1080p), 3, 4
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_7; mkdir -p $O; export TMPDIR=/tmp
cd $R
b() { # name, extra args
  python bench.py $2 --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1:", round(d["value"],1), "pairs/s  frac", round(d["roofline"]["frac"],3), "noop", d["config"].get("noop_step_fraction"))
PY
}
for rep in 1 2; do b auto_$rep ""; b g6_$rep "--step-group 6"; b g3_$rep "--step-group 3"; b g4_$rep "--step-group 4"; done
b w224_auto "--width 224 --height 224 --clips 64"
b w224_g6 "--width 224 --height 224 --clips 64 --step-group 6"
