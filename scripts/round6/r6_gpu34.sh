#!/bin/bash
# round 6, GPU call 34: the streaming SOR's early loads re-tuned on its final form (2 / 3 / 4 sweeps before the tile's end)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_34; mkdir -p $O; export TMPDIR=/tmp; cd $R
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for e in 3 2 4; do
  L=""; [ $e != 3 ] && L="DFX_LIBRARY=$R/build/variants/libdfx_e$e.so"
  env $L timeout 600 python bench.py --algo brox --frames 131 $B 2> $O/err_$e.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1080p early $e:', round(d['value'],2), d.get('parity_check',{}).get('max_abs'))"
done; done
