#!/bin/bash
# round 3, GPU session 2: parity of the pruned / new kernels, TVL1 fast-vs-exact data, Brox r2-vs-r3, Farneback PCIe timeline
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-others --no-pcie"
for m in exact fast exact fast; do timeout 300 $B --steps 3 --math $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tvl1 1080p', '$m', round(d['value'],1), 'pairs/s, iters', round(d['config']['mean_inner_iterations_per_pair'],1), 'step us', round(d['roofline']['avg_launch_us'],1))"; done | tee $O/tvl1_math_ab.txt
for v in 0 16 0 16; do timeout 300 $B --algo brox --steps 2 --variant $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('brox 1080p variant', $v, round(d['value'],2), 'pairs/s')"; done | tee $O/brox_ab.txt
for v in 0 16; do timeout 300 $B --algo brox --width 3840 --height 2160 --frames 34 --step 2 --steps 2 --variant $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('brox 4K s2 variant', $v, round(d['value'],2), 'pairs/s')"; done | tee -a $O/brox_ab.txt
timeout 600 python scripts/tvl1_fast_eval.py $O/tvl1_fast_vs_exact.md > $O/tvl1_fast_eval.log 2>&1; tail -6 $O/tvl1_fast_eval.log
# Farneback f32-out through the host-pointer entry points: where does the time go?  (kernel + memory-copy timeline)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/farn_pcie_trace -- python $GRAFT_REPO_ROOT/bench.py --algo farn --steps 1 --warmup 1 --no-cpu-baseline --no-others > $GRAFT_REPO_ROOT/$O/farn_pcie_bench.json 2> $GRAFT_REPO_ROOT/$O/farn_pcie.err ); echo "trace rc=$?"
find $O/farn_pcie_trace -name "*.csv" | head; du -sh $O
