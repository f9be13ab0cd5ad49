#!/bin/bash
# round 6, GPU call 42: fuse_k (inner iterations per step launch) re-checked on the final kernels: 3 / 4 (default) / 5 / 6
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_42; mkdir -p $O; export TMPDIR=/tmp; cd $R
B="--steps 6 --warmup 2 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for k in 4 3 5 6; do
  timeout 600 python bench.py --fuse-k $k $B 2> $O/err_$k.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tvl1 1080p fuse_k $k:', round(d['value'],2), d.get('parity_check',{}).get('max_abs'), d['roofline'].get('useful_frac'))"
done; done
