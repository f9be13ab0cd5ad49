#!/bin/bash
# round 3, GPU session 11: which engine runs device-to-host copies (shader blit or SDMA) for each allocation / API variant, and
# whether the runtime's environment switches change it; parity of the packed Brox stage 1
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
count() { # dir -> "blit kernels N, SDMA D2H records M"
  k=$(find $1 -name "*kernel_trace.csv" | head -1); m=$(find $1 -name "*memory_copy_trace.csv" | head -1)
  nb=0; ns=0
  [ -n "$k" ] && nb=$(grep -c copyBuffer $k)
  [ -n "$m" ] && ns=$(grep -c DEVICE_TO_HOST $m)
  echo "blit kernel dispatches $nb, SDMA device-to-host records $ns"
}
for v in 0 1 2 3 4 5 6 7 8; do
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/t$v -- $R/build/d2h_engine_probe $v > $R/$O/t$v.log 2>&1 )
  echo "$(grep variant $O/t$v.log) | $(count $O/t$v)"; rm -rf $O/t$v
done | tee $O/d2h_engine.txt
for e in "HSA_ENABLE_SDMA=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_FORCE_SDMA_SIZE=1" "HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE=1" "HSA_REV_COPY_DIR=1" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2"; do
  ( cd /tmp && env $e timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/te -- $R/build/d2h_engine_probe 0 > $R/$O/te.log 2>&1 )
  echo "$e : $(grep variant $O/te.log) | $(count $O/te)"; rm -rf $O/te
done | tee -a $O/d2h_engine.txt
timeout 600 python -m pytest tests/test_brox_gpu.py -m gpu -x -q > $O/pytest_brox.log 2>&1; echo "brox pytest rc=$?"; tail -2 $O/pytest_brox.log
for rep in 1 2; do timeout 300 python bench.py --algo brox --steps 2 --no-cpu-baseline --no-others --no-pcie 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('brox 1080p', round(d['value'],1))"; done | tee $O/brox_rate.txt
