#!/bin/bash
# round 6, GPU call 41: the warp-and-head kernel with (I1, I1x) side by side in its LDS image tile (one ds_read_b64 per tap for both) against the tree before
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_41; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests/test_tvl1_gpu.py tests/test_content_classes_gpu.py tests/test_edge_sizes_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2 3; do
for v in new base; do
  L=""; [ $v = base ] && L="DFX_LIBRARY=$R/build/variants/libdfx_base.so"
  env $L timeout 600 python bench.py $B 2> $O/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tvl1 1080p $v:', round(d['value'],2), d.get('parity_check',{}).get('max_abs'), d['roofline'].get('avg_launch_us'))"
done; done
for v in new base; do
  L=""; [ $v = base ] && L="DFX_LIBRARY=$R/build/variants/libdfx_base.so"
  env $L timeout 600 python bench.py --width 224 --height 224 --clips 64 --steps 4 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity 2> $O/err_224_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tvl1 224x64 $v:', round(d['value'],1))"
  env $L timeout 600 python bench.py --clip hard --frames 66 --steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc 2> $O/err_hard_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tvl1 hard $v:', round(d['value'],2), d.get('parity_check',{}).get('max_abs'))"
done
