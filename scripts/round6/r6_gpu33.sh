#!/bin/bash
# round 6, GPU call 33: which side of the buffer addressing breaks TVL1 — loads only (L), stores only (S), the head kernel only (H)
set -u
R=$GRAFT_REPO_ROOT; cd $R
for v in L S H; do
  echo "== $v"
  DFX_LIBRARY=$R/build/variants/libdfx_$v.so timeout 300 python -m pytest tests/test_tvl1_gpu.py -m gpu -x -q -k "single_pair and libdevice" 2>&1 | tail -2
done
