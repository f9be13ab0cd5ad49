#!/bin/bash
# round 6, GPU call 9: the whole GPU suite on the current tree + the default bench line (valu_frac normalised per XCD)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_9; mkdir -p $O; export TMPDIR=/tmp
cd $R; make -s host > $O/make_host.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; grep "leg \|PMC" $O/bench_default.err; wc -c $O/bench_default.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_9/bench_default.json").read().strip().splitlines()[-1])
print("tvl1", d["value"], {k:v for k,v in d["roofline"].items() if k not in ("kernel","limiter","traffic_source","valu_source")})
print(d["roofline"].get("valu_source")); print(d.get("parity_check"))
c=d["config"]; print({k:v for k,v in c.items() if not isinstance(v,(dict,list,str))})
PY
