#!/bin/bash
# round 3, GPU session 12: what makes the runtime execute a device-to-host copy as a shader blit — a cross-stream event
# dependency, concurrent kernels, the stream, or the HIP runtime torch bundles (the library's copies inside a Python process)
O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
count() { k=$(find $1 -name "*kernel_trace.csv" | head -1); m=$(find $1 -name "*memory_copy_trace.csv" | head -1); nb=0; ns=0
  [ -n "$k" ] && nb=$(grep -c copyBuffer $k); [ -n "$m" ] && ns=$(grep -c DEVICE_TO_HOST $m)
  echo "blit kernel dispatches $nb, SDMA device-to-host records $ns"; }
for v in 0 9 10 11; do
  echo "no profiler: $($R/build/d2h_engine_probe $v 2>&1 | grep variant)"
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/t$v -- $R/build/d2h_engine_probe $v > $R/$O/t$v.log 2>&1 )
  echo "$(grep variant $O/t$v.log) | $(count $O/t$v)"; rm -rf $O/t$v
done | tee $O/d2h_engine_dependency.txt
# the library inside a Python process (torch's bundled HIP runtime), Farneback float flows out, one pass
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/py -- python $R/scripts/pcie_path_probe.py farn 0 0 1920 1080 300 2 > $R/$O/py.log 2>&1 )
echo "library in python: $(grep pass $O/py.log | tail -1) | $(count $O/py)" | tee -a $O/d2h_engine_dependency.txt
python - <<'PY' | tee -a gpurun_out/r3l/d2h_engine_dependency.txt
import ctypes, os
import torch
print("torch", torch.__version__, "hip", torch.version.hip)
for l in open("/proc/self/maps"):
    if "libamdhip64" in l or "libhsa-runtime" in l:
        print(l.split()[-1]); 
PY
rm -rf $O/py
# the same without the (redundant) cross-stream event wait in front of the downloads
( cd /tmp && DFX_LIBRARY=$R/build/variants/libdfx_noevwait.so timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/py2 -- python $R/scripts/pcie_path_probe.py farn 0 0 1920 1080 300 2 > $R/$O/py2.log 2>&1 )
echo "library in python, no event wait: $(grep pass $O/py2.log | tail -1) | $(count $O/py2)" | tee -a $O/d2h_engine_dependency.txt; rm -rf $O/py2
echo "rates without the profiler:" | tee -a $O/d2h_engine_dependency.txt
( python scripts/pcie_path_probe.py farn 0 0 2>/dev/null; DFX_LIBRARY=build/variants/libdfx_noevwait.so python scripts/pcie_path_probe.py farn 0 0 2>/dev/null ) | tee -a $O/d2h_engine_dependency.txt
