#!/bin/bash
# round 6, GPU call 19: streaming Brox SOR, LDS-DMA with the nt cache policy (aux = 2) against the default policy
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_19; mkdir -p $O; cd $R
b() { env $2 python bench.py --algo brox $3 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1:", round(d["value"],2), "pairs/s  launch us", round(d["roofline"]["avg_launch_us"],1), "parity", d.get("parity_check",{}).get("max_abs"))
PY
}
for rep in 1 2; do b def_$rep X=1 "--frames 131"; b nt_$rep DFX_LIBRARY=$R/build/variants/libdfx_nt.so "--frames 131"; done
b 4k_def X=1 "--width 3840 --height 2160 --frames 66 --step 2"; b 4k_nt DFX_LIBRARY=$R/build/variants/libdfx_nt.so "--width 3840 --height 2160 --frames 66 --step 2"
