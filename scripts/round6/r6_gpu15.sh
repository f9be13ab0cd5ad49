#!/bin/bash
# round 6, GPU call 16: + the next tile u, v, du, dv loads issued before the stores; streaming only where a CU gets >= 4 tiles (persistent workgroups, LDS-DMA prefetch of the next tile's coefficient planes)
# against one workgroup per tile — parity, then rates at 1080p and 4K -s=2
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_16; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_brox_gpu.py tests/test_content_classes_gpu.py tests/test_edge_sizes_gpu.py tests/test_bench_shaped_batch_gpu.py tests/test_png_planes_gpu.py -q -m gpu -x -k "brox" 2>&1 | tail -6 > $O/pytest.log
cat $O/pytest.log
b() { python bench.py --algo brox $2 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1:", round(d["value"],2), "pairs/s  frac", round(d["roofline"]["frac"],3), "launch us", round(d["roofline"]["avg_launch_us"],1), "parity", d.get("parity_check",{}).get("max_abs"))
PY
}
for rep in 1 2; do b stream_$rep "--frames 131"; b pertile_$rep "--frames 131 --variant 256"; done
b 4k_stream "--width 3840 --height 2160 --frames 66 --step 2"; b 4k_pertile "--width 3840 --height 2160 --frames 66 --step 2 --variant 256"
