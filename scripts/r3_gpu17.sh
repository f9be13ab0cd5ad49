#!/bin/bash
# round 3, GPU session 17: the JPEG encoders with libjpeg's integer transform (files = libjpeg-turbo's bytes): device vs Pillow /
# goldens / host encoder, the shell's CLI tests, and the JPEG-out rates
O=gpurun_out/r3q; mkdir -p $O
timeout 1200 python -m pytest tests/test_jpeg_gpu.py tests/test_jpeg_libjpeg_pin.py tests/test_host_shell.py tests/test_quant_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-others > $O/bench_tvl1.json 2> $O/bench_tvl1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3q/bench_tvl1.json').read().strip().splitlines()[-1]); p=d['config']['pcie_inclusive']
print('tvl1 resident', round(d['value'],1), 'f32', round(p['value'],1), 'u8', round(p['u8_bounded_planes_out'],1), 'jpeg', round(p['jpeg_files_out']['value'],1), 'mean file bytes', round(p['jpeg_files_out']['mean_file_bytes']))
PY
