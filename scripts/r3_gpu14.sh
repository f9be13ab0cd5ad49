#!/bin/bash
# round 3, GPU session 14: Brox SOR with the LDS tile split by column parity (no stride-2 bank conflicts); JPEG goldens; smoke
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_brox_gpu.py tests/test_jpeg_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
for v in 0 16 0 16; do timeout 300 python bench.py --algo brox --steps 2 --no-cpu-baseline --no-others --no-pcie --variant $v 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('brox 1080p variant', $v, round(d['value'],1))"; done | tee $O/brox_ab.txt
timeout 300 python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 --steps 2 --no-cpu-baseline --no-others --no-pcie 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('brox 4K s2', round(d['value'],2))" | tee -a $O/brox_ab.txt
