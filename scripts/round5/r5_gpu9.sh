#!/bin/bash
# round 5, GPU call 9: LDS-tiled warp: strip height and waves per SIMD (A/B builds), against the gather kernel
O=gpurun_out/r5_9; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
run() { tag=$1; lib=$2; var=$3
  DFX_LIBRARY=$lib python bench.py --variant $var --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/b_$tag.json 2> $O/b_$tag.err
  python - <<PY
import json
a=json.loads(open("gpurun_out/r5_9/b_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(a["value"],1), "launch us", round(a["roofline"]["avg_launch_us"],1))
PY
}
D=/root/repo/denseflow_amd/lib/libdfx.so; V=/root/repo/build/variants
for r in 1 2; do
  run gather $D 0
  run lds_sr16_wps4 $D 32
  run lds_sr16_wps5 $V/libdfx_warp_sr16_wps5.so 32
  run lds_sr32_wps4 $V/libdfx_warp_sr32_wps4.so 32
  run lds_sr32_wps3 $V/libdfx_warp_sr32_wps3.so 32
done
DFX_LIBRARY=$V/libdfx_warp_sr32_wps4.so timeout 300 python -m pytest tests/test_tvl1_gpu.py -m gpu -q -x -k "warp_through" 2>&1 | tail -2
