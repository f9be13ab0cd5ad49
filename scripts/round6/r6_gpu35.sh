#!/bin/bash
# round 6, GPU call 35: the streaming SOR with 44 of 64 rows by DMA (measurement build, wrong flows): what a row ring down a tile column could buy
# (measurement builds; the flows are wrong, so only the SOR kernel's own time is compared)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_35; mkdir -p $O; export TMPDIR=/tmp; cd $R
for v in real dbg7; do
  L=""; [ $v != real ] && L="DFX_LIBRARY=$R/build/variants/libdfx_$v.so"
  ( cd /tmp && env $L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o p -- python $R/bench.py --algo brox --frames 131 --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/profiled_$v.json 2> $O/stats_$v.err
  find $O/stats_$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$v.csv \; ; rm -rf $O/stats_$v
  echo "== $v"; python scripts/kstats.py $O/kernel_stats_$v.csv | head -1
done
