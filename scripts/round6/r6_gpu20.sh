#!/bin/bash
# round 6, GPU call 20: Brox prolongation with the separable bicubic weights formed once per pixel — parity, rate, kernel stats
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_20; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests/test_brox_gpu.py tests/test_content_classes_gpu.py tests/test_edge_sizes_gpu.py tests/test_bench_shaped_batch_gpu.py -q -m gpu -x -k "brox" 2>&1 | tail -3
for rep in 1 2; do
python bench.py --algo brox --frames 131 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/bench_$rep.json 2> $O/bench_$rep.err
python - <<PY
import json
d=json.loads(open("$O/bench_$rep.json").read().strip().splitlines()[-1])
print("1080p rep $rep:", round(d["value"],2), "pairs/s parity", d.get("parity_check",{}).get("max_abs"))
PY
done
python bench.py --algo brox --width 3840 --height 2160 --frames 66 --step 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/bench_4k.json 2> $O/bench_4k.err
python - <<PY
import json
d=json.loads(open("$O/bench_4k.json").read().strip().splitlines()[-1])
print("4k:", round(d["value"],2), "pairs/s parity", d.get("parity_check",{}).get("max_abs"))
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/bench.py --algo brox --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/profiled.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/brox_kernel_stats.csv \; ; rm -rf $O/stats
python scripts/kstats.py $O/brox_kernel_stats.csv | head -9
