#!/bin/bash
# round 6, GPU call 30: streaming SOR — tiles numbered block by block (8 x 4, 4 x 8, 6 x 6) against row by row (the b1 build)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_30; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_brox_gpu.py -m gpu -x -q > $O/pytest_brox.log 2>&1; tail -2 $O/pytest_brox.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for v in b84 b48 b66 b1; do
  L=""; [ $v != b84 ] && L="DFX_LIBRARY=$R/build/variants/libdfx_$v.so"
  env $L timeout 600 python bench.py --algo brox --frames 131 $B 2> $O/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1080p $v:', d['value'], d.get('parity_check',{}).get('max_abs'))"
done; done
for v in b84 b1; do
  L=""; [ $v != b84 ] && L="DFX_LIBRARY=$R/build/variants/libdfx_$v.so"
  env $L timeout 600 python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 $B 2> $O/err_4k_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('4k $v:', d['value'], d.get('parity_check',{}).get('max_abs'))"
done
