#!/bin/bash
# round 6, GPU call 24: streaming SOR — early loads 3 / 4 / 5 sweeps before the end; cycle counts of the phases (debug builds)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_24; mkdir -p $O; export TMPDIR=/tmp; cd $R
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for e in 3 4 5; do
  env DFX_LIBRARY=$R/build/variants/libdfx_e$e.so timeout 600 python bench.py --algo brox --frames 131 $B 2> $O/err_$e.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1080p early $e:', d['value'], d.get('parity_check',{}).get('max_abs'))"
done; done
for e in 0 3; do
  echo "== debug build, early $e"
  env DFX_LIBRARY=$R/build/variants/libdfx_dbg_e$e.so timeout 600 python bench.py --algo brox --frames 66 --steps 1 --warmup 0 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity 2>&1 | grep "sor_stream wg" | head -8
done
