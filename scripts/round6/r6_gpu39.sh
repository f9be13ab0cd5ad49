#!/bin/bash
# round 6, GPU call 39: the level context from the kernel-argument segment on the once-per-level paths only (finish_level, the
# in-kernel warp phase): step kernel only / both kernels / the tree before
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_39; mkdir -p $O; export TMPDIR=/tmp; cd $R
DFX_LIBRARY=$R/build/variants/libdfx_both.so timeout 900 python -m pytest tests/test_tvl1_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2 3; do
for v in steponly both base; do
  env DFX_LIBRARY=$R/build/variants/libdfx_$v.so timeout 600 python bench.py $B 2> $O/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tvl1 1080p $v:', round(d['value'],2), d.get('parity_check',{}).get('max_abs'), d['roofline'].get('avg_launch_us'))"
done; done
