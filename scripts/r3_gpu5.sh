#!/bin/bash
# round 3, GPU session 5: the device JPEG encoder
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_jpeg_gpu.py tests/test_jpeg_host.py -x -q > $O/pytest_jpeg.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_jpeg.log
for a in farn tvl1; do timeout 400 python bench.py --no-cpu-baseline --no-others --steps 2 --algo $a 2>$O/bench_$a.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['config']['pcie_inclusive']
print('$a resident', round(d['value'],1), 'f32', round(p['value'],1), 'u8', round(p['u8_bounded_planes_out'],1), 'jpeg', round(p['jpeg_files_out']['value'],1), 'mean file bytes', round(p['jpeg_files_out']['mean_file_bytes']))"; done | tee $O/jpeg_rates.txt
tail -3 $O/bench_farn.err
