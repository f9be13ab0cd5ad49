#!/bin/bash
# round 3, GPU session 21: kernel statistics of the 224x224 workload, one clip per FlowBuffer and 16 clips joined
O=gpurun_out/r3u; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 1 16; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_$n -- python $R/bench.py --width 224 --height 224 --clips $n --steps 3 --no-cpu-baseline --no-others --no-pcie > $R/$O/bench_224_${n}clips.json 2>/dev/null )
  f=$(find $O/stats_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_tvl1_224x224_${n}clips_kernel_stats.csv && echo "== $n clip(s) per FlowBuffer" && python scripts/kstats.py $f | head -8; rm -rf $O/stats_$n
done
