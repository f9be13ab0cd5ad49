#!/bin/bash
# round 5, GPU call 6: Brox SOR guarded LDS tile A/B (the call-5 loop passed an empty DFX_LIBRARY for the default build)
O=gpurun_out/r5_6; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
for r in 1 2 3; do for v in guard noguard; do
  L=/root/repo/denseflow_amd/lib/libdfx.so; [ $v = noguard ] && L=/root/repo/build/variants/libdfx_brox_noguard.so
  DFX_LIBRARY=$L python bench.py --algo brox --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_brox_$v.json 2> $O/bench_brox_$v.err
  DFX_LIBRARY=$L python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_brox4k_$v.json 2>> $O/bench_brox_$v.err
  python - <<PY
import json
a=json.loads(open("gpurun_out/r5_6/bench_brox_$v.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/r5_6/bench_brox4k_$v.json").read().strip().splitlines()[-1])
print("$v 1080p", round(a["value"],1), "launch us", round(a["roofline"]["avg_launch_us"],1), " 4K -s=2", round(b["value"],2))
PY
done; done
