#!/bin/bash
# Round 4, closing run on the final tree: GPU suite, smoke, the default bench line, rocprofv3 --kernel-trace --stats of
# bench.py per algorithm (the summaries roofline.avg_launch_us must agree with), the other configurations' own lines.
O=gpurun_out/r4_final; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo; R=/root/repo
timeout 1400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_default.json)"
for a in tvl1 farn brox; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_$a -o p -- python $R/bench.py --algo $a --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc ) > $O/bench_${a}_1080p_profiled.json 2> $O/stats_$a.err
  find $O/stats_$a -name "*kernel_stats.csv" -exec cp {} $O/bench_${a}_1080p_kernel_stats.csv \;
  rm -rf $O/stats_$a
  head -4 $O/bench_${a}_1080p_kernel_stats.csv | cut -c1-150
done
python bench.py --algo farn --no-others --no-cpu-baseline > $O/bench_farn_1080p.json 2>> $O/err.log
python bench.py --algo brox --no-others --no-cpu-baseline --no-live-pmc > $O/bench_brox_1080p.json 2>> $O/err.log
python - <<'PY'
import json
O="gpurun_out/r4_final"
d=json.load(open(O+"/bench_default.json"))
print("tvl1", d["value"], d["roofline"]["frac"], d["roofline"]["traffic_frac"], d["roofline"]["avg_launch_us"])
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
for a in ("farn","brox"):
    x=json.load(open(f"{O}/bench_{a}_1080p.json")); print(a, x["value"], x["roofline"]["frac"], x["roofline"]["traffic_frac"], x["roofline"]["avg_launch_us"], x["config"].get("pcie_f32_pairs_per_s"), x["config"].get("pcie_jpeg_pairs_per_s"))
PY
