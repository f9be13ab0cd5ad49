#!/bin/bash
# Round 4: the whole default bench line (live PMC legs included), timed
O=gpurun_out/r4_bench2; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
t0=$(date +%s.%N)
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s"
tail -5 $O/bench_default.err
wc -c $O/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench2/bench_default.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:1200])
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
print(d["cpu_baseline"])
PY
