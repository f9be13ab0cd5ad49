#!/bin/bash
# round 6, GPU call 26: streaming SOR — early loads (3) confirmed on the tests; wave issue priorities as an experiment
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_26; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_brox_gpu.py tests/test_content_classes_gpu.py -m gpu -x -q -k "brox or Brox" > $O/pytest_brox.log 2>&1; tail -2 $O/pytest_brox.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for e in 0 1 2; do
  L=""; [ $e != 0 ] && L="DFX_LIBRARY=$R/build/variants/libdfx_prio$e.so"
  env $L timeout 600 python bench.py --algo brox --frames 131 $B 2> $O/err_$e.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1080p prio $e:', d['value'], d.get('parity_check',{}).get('max_abs'))"
done; done
timeout 600 python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 $B 2> $O/err_4k.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('4k:', d['value'], d.get('parity_check',{}).get('max_abs'))"
