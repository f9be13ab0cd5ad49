#!/bin/bash
# round 3, GPU session 13: the PCIe-inclusive path on the system HIP runtime (a torch-free process), which copies device-to-host with SDMA
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
python scripts/pcie_path_probe_notorch.py make 1920 1080 300 /tmp/clip1080.npy
python scripts/pcie_path_probe_notorch.py make 224 224 300 /tmp/clip224.npy
( DFX_NO_TORCH=1 timeout 300 python scripts/pcie_path_probe_notorch.py run farn /tmp/clip1080.npy
  DFX_NO_TORCH=1 timeout 300 python scripts/pcie_path_probe_notorch.py run tvl1 /tmp/clip1080.npy 2
  DFX_NO_TORCH=1 timeout 300 python scripts/pcie_path_probe_notorch.py run tvl1 /tmp/clip224.npy
  DFX_NO_TORCH=1 timeout 300 python scripts/pcie_path_probe_notorch.py run brox /tmp/clip1080.npy 1 ) 2>&1 | grep -v amdgpu.ids | tee $O/pcie_system_runtime.txt
timeout 300 python bench.py --algo farn --steps 3 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['config']['pcie_inclusive']
print('same box, torch runtime: farn resident', round(d['value'],1), 'f32', round(p['value'],1), 'u8', round(p['u8_bounded_planes_out'],1), 'jpeg', round(p['jpeg_files_out']['value'],1))" | tee -a $O/pcie_system_runtime.txt
