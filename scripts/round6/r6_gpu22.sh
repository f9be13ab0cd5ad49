#!/bin/bash
# round 6, GPU call 22: the streaming SOR that never waits for its own stores (fixed store count + vmcnt(4), DMA by hand)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_22; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_brox_gpu.py tests/test_content_classes_gpu.py -m gpu -x -q -k "brox or Brox" > $O/pytest_brox.log 2>&1; tail -3 $O/pytest_brox.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for v in 0 256; do
  timeout 600 python bench.py --algo brox --frames 131 --variant $v $B 2> $O/err_1080_${v}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1080p variant $v:', d['value'], d.get('parity_check',{}).get('max_abs'))"
done; done
for v in 0 256; do
  timeout 600 python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 --variant $v $B 2> $O/err_4k_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('4k variant $v:', d['value'], d.get('parity_check',{}).get('max_abs'))" || true
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/bench.py --algo brox --frames 131 --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/profiled.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
python scripts/kstats.py $O/kernel_stats.csv | head -6
