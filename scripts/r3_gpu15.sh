#!/bin/bash
# round 3, GPU session 15: what the Brox SOR kernel's time consists of — builds without the sweeps / without the load traffic
O=gpurun_out/r3o; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" build/variants/libdfx_sor_nosweeps.so build/variants/libdfx_sor_noloads.so; do
  ( cd /tmp && DFX_LIBRARY=${lib:+$R/$lib} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/st -- python $R/bench.py --algo brox --steps 1 --warmup 1 --frames 130 --no-cpu-baseline --no-others --no-pcie > /dev/null 2>&1 )
  f=$(find $O/st -name "*kernel_stats.csv" | head -1); echo "== ${lib:-default}"; python scripts/kstats.py $f | head -3; rm -rf $O/st
done | tee $O/brox_sor_decomposition.txt
