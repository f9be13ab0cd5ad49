#!/bin/bash
# round 6, GPU call 36: the streaming SOR's coefficient planes as a row ring down tile columns (44 instead of 64 rows by DMA
# for every tile but a unit's first) against the tree before
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_36; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests/test_brox_gpu.py tests/test_content_classes_gpu.py -m gpu -x -q -k "brox or Brox" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for v in new base; do
  L=""; [ $v = base ] && L="DFX_LIBRARY=$R/build/variants/libdfx_base.so"
  env $L timeout 600 python bench.py --algo brox --frames 131 $B 2> $O/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('brox 1080p $v:', round(d['value'],2), d.get('parity_check',{}).get('max_abs'))"
done; done
for v in new base; do
  L=""; [ $v = base ] && L="DFX_LIBRARY=$R/build/variants/libdfx_base.so"
  env $L timeout 600 python bench.py --algo brox --width 3840 --height 2160 --frames 66 --step 2 $B 2> $O/err_4k_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('brox 4k $v:', round(d['value'],2), d.get('parity_check',{}).get('max_abs'))"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/bench.py --algo brox --frames 131 --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/profiled.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
python scripts/kstats.py $O/kernel_stats.csv | head -3
