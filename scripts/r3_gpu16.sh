#!/bin/bash
# round 3, GPU session 16: Brox SOR with in-patch neighbours in registers and {u+du, v+dv} as one 8-byte LDS entry, against the
# parity-split build of the commit before (build/variants/libdfx_parity_split.so), alternating on one box
O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_brox_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
P=$PWD/build/variants/libdfx_parity_split.so
for lib in "" $P "" $P; do
  if [ -n "$lib" ]; then export DFX_LIBRARY=$lib; n=parity_split_b32; else unset DFX_LIBRARY; n=own_regs_b64; fi
  timeout 300 python bench.py --algo brox --steps 2 --no-cpu-baseline --no-others --no-pcie 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('brox 1080p', '$n', round(d['value'],1))"; done | tee $O/brox_ab.txt
unset DFX_LIBRARY
timeout 300 python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 --steps 2 --no-cpu-baseline --no-others --no-pcie 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('brox 4K s2 own_regs_b64', round(d['value'],2))" | tee -a $O/brox_ab.txt
