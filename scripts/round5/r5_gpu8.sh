#!/bin/bash
# round 5, GPU call 8: the backward warp through an LDS tile (DFX_VAR_TVL1_WARP_LDS = 32): same bits? faster?
O=gpurun_out/r5_8; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 600 python -m pytest tests/test_tvl1_gpu.py -m gpu -q -x -k "warp_through or tile_geometry" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for r in 1 2; do for v in 0 32; do
  python bench.py --variant $v --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_v$v.json 2> $O/bench_v$v.err
  python bench.py --variant $v --clip hard --frames 66 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/hard_v$v.json 2>> $O/bench_v$v.err
  python - <<PY
import json
a=json.loads(open("gpurun_out/r5_8/bench_v$v.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/r5_8/hard_v$v.json").read().strip().splitlines()[-1])
print("variant $v headline", round(a["value"],1), "launch us", round(a["roofline"]["avg_launch_us"],1), " hard", round(b["value"],1))
PY
done; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o p -- python /root/repo/bench.py --variant 32 --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/prof.json 2> $O/prof.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_warp_lds.csv \; ; rm -rf $O/stats; head -3 $O/kernel_stats_warp_lds.csv | cut -c1-140
