#!/bin/bash
# round 3, GPU session 6: whole GPU suite on the current tree, end-to-end shell rates with the device JPEG encoder, the full bench line
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 600 python scripts/e2e_cli_rate.py 1920 1080 513 2>&1 | grep -v amdgpu.ids | tee $O/e2e_1080p.log
timeout 600 python scripts/e2e_cli_rate.py 224 224 300 32 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x32clips.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3f/bench_default.json').read().strip().splitlines()[-1])
p=d['config']['pcie_inclusive']
print('tvl1 resident', round(d['value'],1), 'f32', round(p['value'],1), 'u8', round(p['u8_bounded_planes_out'],1), 'jpeg', round(p['jpeg_files_out']['value'],1), 'in-flight u8/jpeg', round(p['flowbuffers_in_flight']['u8_bounded_planes_out'],1), round(p['flowbuffers_in_flight']['jpeg_files_out'],1))
for w in d['config']['other_workloads']:
    q=w.get('pcie_inclusive',{})
    print(w['workload'][:40], round(w.get('pairs_per_s',0),1), 'f32', round(q.get('value',0),1), 'jpeg', round(q.get('jpeg_files_out',{}).get('value',0),1), 'in-flight jpeg', round(q.get('flowbuffers_in_flight',{}).get('jpeg_files_out',0),1), w.get('error',''))
print('cpu', d['cpu_baseline']['value'])
PY
