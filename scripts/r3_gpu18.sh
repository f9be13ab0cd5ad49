#!/bin/bash
# round 3, GPU session 18: several short clips joined into one FlowBuffer (dfx_next_segments) — parity tests, and the 224x224 rate
# against the number of clips per FlowBuffer (BASELINE configs[3]: a list of 224x224x300 clips)
O=gpurun_out/r3r; mkdir -p $O
timeout 1200 python -m pytest tests/test_segments_gpu.py tests/test_tvl1_gpu.py tests/test_async_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for n in 1 2 4 8 16 32; do
  timeout 300 python bench.py --width 224 --height 224 --clips $n --steps 3 --no-cpu-baseline --no-others --no-pcie 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('tvl1 224x224 clips/FlowBuffer', $n, 'pairs/s', round(d['value'],1), 'pairs/launch', c['pairs_per_launch'], 'launches/pair', round(c['kernel_launches_per_pair'],3), 'noop', round(c['noop_step_fraction'],3), 'frac', round(d['roofline']['frac'],3))"
done | tee $O/tvl1_224_clips_per_flowbuffer.txt
for a in farn brox; do for n in 1 16; do
  timeout 300 python bench.py --algo $a --width 224 --height 224 --clips $n --steps 2 --no-cpu-baseline --no-others --no-pcie 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a 224x224 clips/FlowBuffer', $n, 'pairs/s', round(d['value'],1))"
done; done | tee -a $O/tvl1_224_clips_per_flowbuffer.txt
timeout 300 python bench.py --width 224 --height 224 --clips 16 --steps 2 --no-cpu-baseline --no-others > $O/bench_tvl1_224x224_16clips.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3r/bench_tvl1_224x224_16clips.json').read().strip().splitlines()[-1]); p=d['config']['pcie_inclusive']
print('16 clips: resident', round(d['value'],1), 'f32', round(p['value'],1), 'u8', round(p['u8_bounded_planes_out'],1), 'jpeg', round(p['jpeg_files_out']['value'],1), 'in flight u8/jpeg', round(p['flowbuffers_in_flight']['u8_bounded_planes_out'],1), round(p['flowbuffers_in_flight']['jpeg_files_out'],1), 'identical', p['flowbuffers_in_flight']['outputs_identical'])
PY
