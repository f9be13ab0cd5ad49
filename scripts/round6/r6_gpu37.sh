#!/bin/bash
# round 6, GPU call 37: why the row ring buys nothing — units of 1 tile (the old order), of 3, of 5 without the ring, of 5 with it
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_37; mkdir -p $O; export TMPDIR=/tmp; cd $R
for v in new seg1 seg3 ringoff; do
  L=""; [ $v != new ] && L="DFX_LIBRARY=$R/build/variants/libdfx_$v.so"
  ( cd /tmp && env $L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o p -- python $R/bench.py --algo brox --frames 131 --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/profiled_$v.json 2> $O/stats_$v.err
  find $O/stats_$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$v.csv \; ; rm -rf $O/stats_$v
  echo "== $v"; python scripts/kstats.py $O/kernel_stats_$v.csv | head -1
done
