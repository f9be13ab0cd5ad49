#!/bin/bash
# Round 4: whole GPU suite + the default bench line on the tree with the row-stream Farneback kernel
O=gpurun_out/r4_full1; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 1400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_full1/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_frac"])
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
PY
