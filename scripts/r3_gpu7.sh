#!/bin/bash
# round 3, GPU session 7: Farneback in Infinity-Cache-sized batches; kernel breakdown of Brox with the new SOR; shell test
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_host_shell.py tests/test_brox_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
rate() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'pairs/s, step us', round(d['roofline']['avg_launch_us'],1), 'launches/pair', round(d['config']['kernel_launches_per_pair'],1))"; }
B="python bench.py --no-cpu-baseline --no-others --no-pcie --steps 2 --algo farn"
for b in 1 2 4 8 16 32 0; do timeout 300 $B --max-batch $b 2>/dev/null | rate "farn max-batch $b"; done | tee $O/farn_batch.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/brox_stats -- python $GRAFT_REPO_ROOT/bench.py --algo brox --steps 2 --no-cpu-baseline --no-others --no-pcie > $GRAFT_REPO_ROOT/$O/brox_bench.json 2>/dev/null )
f=$(find $O/brox_stats -name "*kernel_stats.csv" | head -1); python scripts/kstats.py $f | tee $O/brox_kernel_stats.txt; cp $f $O/bench_brox_1080p_kernel_stats.csv; rm -rf $O/brox_stats
