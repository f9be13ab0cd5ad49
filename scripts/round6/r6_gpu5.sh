#!/bin/bash
# round 6, GPU call 6: head kernel (compiler-scheduled taps), p of a level s first warp neither zeroed nor read — parity, A/B against the two-launch form, timeline
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_6; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_tvl1_gpu.py tests/test_content_classes_gpu.py -q -m gpu -x -k "tvl1 or content" 2>&1 | tail -6 > $O/pytest.log
cat $O/pytest.log
b() { # name, variant
  python bench.py --variant $2 --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1:", round(d["value"],1), "pairs/s  frac", round(d["roofline"]["frac"],3), "parity", d.get("parity_check",{}).get("max_abs"))
PY
}
for rep in 1 2 3; do b head_$rep 0; b nohead_$rep 64; done
cd /tmp
( SWEEP="0:4:0:0" timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_0 -o t -- python $R/scripts/sweep_tvl1.py 1920 1080 130 ) > $O/trace_0.log 2>&1; echo "trace rc=$?"
F=$(find $O/trace_0 -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && python $R/scripts/tvl1_timeline.py "$F" $O/timeline_v0_dispatches.csv > $O/timeline_v0.md 2>$O/timeline_v0.err; rm -rf $O/trace_0; cat $O/timeline_v0.md
