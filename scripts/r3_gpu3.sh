#!/bin/bash
# round 3, GPU session 3: the device-to-host copy kernel, the Farneback iteration kernel at 4 / 5 workgroups per CU
O=gpurun_out/r3c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['config'].get('pcie_inclusive',{})
print('$1', 'resident', round(d['value'],1), 'f32-out', round(p.get('value',0),1), 'u8-out', round(p.get('u8_bounded_planes_out',0),1), 'in-flight u8', round(p.get('flowbuffers_in_flight',{}).get('u8_bounded_planes_out',0),1), 'step us', round(d['roofline']['avg_launch_us'],1))"; }
B="python bench.py --no-cpu-baseline --no-others --steps 3"
for rep in 1 2; do
  timeout 300 $B --algo farn 2>/dev/null | line "farn wgs5 egress48"
  DFX_LIBRARY=build/variants/libdfx_farn_wgs4.so timeout 300 $B --algo farn 2>/dev/null | line "farn wgs4 egress48"
  timeout 300 $B --algo farn --variant 32 2>/dev/null | line "farn wgs5 hipMemcpy"
done | tee $O/farn_ab.txt
for g in 8 16 32 96 256; do timeout 300 $B --algo farn --egress-wgs $g 2>/dev/null | line "farn egress wgs $g"; done | tee -a $O/farn_ab.txt
timeout 300 $B --algo tvl1 2>/dev/null | line "tvl1 egress48" | tee $O/tvl1_ab.txt
timeout 300 $B --algo tvl1 --variant 32 2>/dev/null | line "tvl1 hipMemcpy" | tee -a $O/tvl1_ab.txt
timeout 300 $B --algo tvl1 --width 224 --height 224 --steps 5 2>/dev/null | line "tvl1 224 egress48" | tee -a $O/tvl1_ab.txt
timeout 300 $B --algo tvl1 --width 224 --height 224 --steps 5 --variant 32 2>/dev/null | line "tvl1 224 hipMemcpy" | tee -a $O/tvl1_ab.txt
