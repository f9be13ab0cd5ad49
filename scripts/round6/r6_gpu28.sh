#!/bin/bash
# round 6, GPU call 28: streaming SOR with (u + du, v + dv) side by side in LDS (one ds_read_b64 per neighbour)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_28; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_brox_gpu.py -m gpu -x -q > $O/pytest_brox.log 2>&1; tail -2 $O/pytest_brox.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
  timeout 600 python bench.py --algo brox --frames 131 $B 2> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1080p:', d['value'], d.get('parity_check',{}).get('max_abs'))"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/bench.py --algo brox --frames 131 --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/profiled.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
python scripts/kstats.py $O/kernel_stats.csv | head -3
