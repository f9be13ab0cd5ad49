#!/bin/bash
# round 3, last GPU session on the final tree (after dfx_next_segments, the shell's joined flow stage and the helper-thread
# gather): GPU suite, smoke, the driver-shaped bench line, the CLI on lists of 224x224 clips
O=gpurun_out/final3; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python bench.py > $O/bench_tvl1_1080p.json 2> $O/bench_tvl1_1080p.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final3/bench_tvl1_1080p.json').read().strip().splitlines()[-1]); p=d['config']['pcie_inclusive']
print('tvl1 1080p', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'f32', round(p['value'],1), 'u8', round(p['u8_bounded_planes_out'],1), 'jpeg', round(p['jpeg_files_out']['value'],1), 'cpu', round(d['cpu_baseline']['value'],3))
for o in d['config']['other_workloads']:
    q=o.get('pcie_inclusive',{})
    print(o['workload'][:70], '|', round(o.get('pairs_per_s',0),1), 'frac', round(o.get('roofline',{}).get('frac',0),3), 'f32', round(q.get('value',0),1), 'jpeg', round(q.get('jpeg_files_out',{}).get('value',0),1), 'in flight jpeg', round(q.get('flowbuffers_in_flight',{}).get('jpeg_files_out',0),1), o.get('error',''))
PY
CONFIGS="device" timeout 300 python scripts/e2e_cli_rate.py 224 224 300 64 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x64clips.log
ALGOS=tvl1 CONFIGS="device" timeout 300 python scripts/e2e_cli_rate.py 224 224 300 192 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x192clips.log
