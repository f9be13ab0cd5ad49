#!/bin/bash
# Round-2 closing run on the final tree: full GPU suite, smoke, the five bench lines (bench.py now also reports the
# FlowBuffers-in-flight leg and per-step TVL1 traffic of both kernels).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final4
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -s KILL 600 python bench.py 2>$O/bench_tvl1.err | tail -1 > $O/bench_tvl1_1080p.json; echo "tvl1 rc=$?"
timeout -s KILL 400 python bench.py --algo farn 2>$O/bench_farn.err | tail -1 > $O/bench_farn_1080p.json; echo "farn rc=$?"
timeout -s KILL 400 python bench.py --algo brox --no-cpu-baseline 2>$O/bench_brox.err | tail -1 > $O/bench_brox_1080p.json; echo "brox rc=$?"
timeout -s KILL 400 python bench.py --algo brox --width 3840 --height 2160 --step 2 --frames 34 --steps 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 > $O/bench_brox_4k_s2.json; echo "brox4k rc=$?"
timeout -s KILL 400 python bench.py --width 224 --height 224 --no-cpu-baseline 2>$O/bench_224.err | tail -1 > $O/bench_tvl1_224x224.json; echo "224 rc=$?"
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; c=d.get("cpu_baseline") or {}; p=d.get("pcie_inclusive") or {}; q=p.get("flowbuffers_in_flight") or {}
    print(d["metric"], round(d["value"],1), "frac", round(r["frac"],3), "traffic_frac", r.get("traffic_frac"), "pcie", p.get("value"), p.get("u8_bounded_planes_out"), "in flight", q.get("u8_bounded_planes_out"), q.get("fraction_of_resident"), q.get("outputs_identical"), "cpu", c.get("value"), c.get("cores"))
except Exception as e:
    print("unparsed", e)
PY
done
tail -3 $O/*.err | cut -c1-300
