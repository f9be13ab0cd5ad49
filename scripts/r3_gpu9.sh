#!/bin/bash
# round 3, GPU session 9: larger FlowBuffers (two full device batches) with a short first one
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_shell.py -m gpu -x -q > $O/pytest_shell.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_shell.log
CONFIGS="device" ALGOS=tvl1,farn timeout 900 python scripts/e2e_cli_rate.py 1920 1080 513 2>&1 | grep -v amdgpu.ids | tee $O/e2e_1080p_513.log
CONFIGS="device" ALGOS=tvl1,farn timeout 900 python scripts/e2e_cli_rate.py 1920 1080 3073 2>&1 | grep -v amdgpu.ids | tee $O/e2e_1080p_3073.log
CONFIGS="device" ALGOS=tvl1,farn timeout 600 python scripts/e2e_cli_rate.py 224 224 300 64 2>&1 | grep -v amdgpu.ids | tee $O/e2e_224x64clips.log
timeout 300 python bench.py --no-cpu-baseline --no-others --steps 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['config']['pcie_inclusive']
print('this box: tvl1 resident', round(d['value'],1), 'jpeg', round(p['jpeg_files_out']['value'],1), 'in flight', round(p['flowbuffers_in_flight']['jpeg_files_out'],1))" | tee $O/box_rate.txt
