#!/bin/bash
# round 3, GPU session 4: which change of the Farneback iteration kernel costs what; what the device-to-host leg does to the compute
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
rate() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'pairs/s, step us', round(d['roofline']['avg_launch_us'],1))"; }
B="python bench.py --no-cpu-baseline --no-others --no-pcie --steps 3 --algo farn"
for rep in 1 2; do
  for v in v0 v1 v2 v4; do DFX_LIBRARY=build/variants/libdfx_farn_$v.so timeout 300 $B 2>/dev/null | rate "farn $v"; done
  timeout 300 $B 2>/dev/null | rate "farn default(vpass1,fetch2=1)"
done | tee $O/farn_kernel_ab.txt
P="python scripts/pcie_path_probe.py farn"
export DFX_LIBRARY=build/variants/libdfx_farn_v0.so
( timeout 200 $P 32 0; timeout 200 $P 0 48; timeout 200 $P 0 8; timeout 200 $P 96 0; timeout 200 $P 64 48; timeout 200 $P 64 16 ) 2>/dev/null | tee $O/farn_d2h_ab.txt
( OUT=u8 timeout 200 $P 32 0; OUT=u8 timeout 200 $P 0 48;  OUT=u8 timeout 200 $P 96 0 ) 2>/dev/null | tee -a $O/farn_d2h_ab.txt
for cfg in "32 0" "0 48" "0 8" "96 0"; do
  n=$(echo $cfg | tr ' ' '_')
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$n -- python $GRAFT_REPO_ROOT/scripts/pcie_path_probe.py farn $cfg 1920 1080 300 2 > $GRAFT_REPO_ROOT/$O/trace_$n.log 2>&1 )
  echo "== variant/egress $cfg"; grep pass $O/trace_$n.log; python scripts/pcie_windows.py $O/trace_$n k_farn
  rm -rf $O/trace_$n
done 2>&1 | tee $O/farn_d2h_windows.txt
