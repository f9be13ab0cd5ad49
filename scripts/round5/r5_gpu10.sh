#!/bin/bash
# round 5, GPU calls 10 / 11: Brox stage 1 — the one-reflection mirror index and the exact 16-instruction 1/sqrt (10), then
# through an LDS tile (11): same bits? faster?  (`old` = the library of the commit before these changes)
O=gpurun_out/r5_10; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_brox_gpu.py tests/test_edge_sizes_gpu.py tests/test_bench_shaped_batch_gpu.py -m gpu -q -x -k "brox or Brox" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2 3; do for v in new old; do
  L=/root/repo/denseflow_amd/lib/libdfx.so; [ $v = old ] && L=/root/repo/build/variants/libdfx_brox_old.so
  DFX_LIBRARY=$L python bench.py --algo brox --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_brox_$v.json 2> $O/bench_brox_$v.err
  DFX_LIBRARY=$L python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_brox4k_$v.json 2>> $O/bench_brox_$v.err
  python - <<PY
import json
a=json.loads(open("gpurun_out/r5_10/bench_brox_$v.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/r5_10/bench_brox4k_$v.json").read().strip().splitlines()[-1])
print("$v 1080p", round(a["value"],1), "launch us", round(a["roofline"]["avg_launch_us"],1), " 4K -s=2", round(b["value"],2))
PY
done; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o p -- python /root/repo/bench.py --algo brox --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/prof.json 2> $O/prof.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_brox.csv \; ; rm -rf $O/stats; head -4 $O/kernel_stats_brox.csv | cut -c1-120
