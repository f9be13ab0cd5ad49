#!/bin/bash
# Round 5, closing run on the final tree: GPU suite, smoke, the default bench line, rocprofv3 --kernel-trace --stats of
# bench.py per algorithm (the summaries roofline.avg_launch_us must agree with), the shell end to end (stages; one pipeline,
# two pipeline threads and two pipeline PROCESSES on the one device).
O=gpurun_out/r5_final; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo; R=/root/repo
make -s host > $O/make_host.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "bytes=$(wc -c < $O/bench_default.json)"
for a in tvl1 farn brox; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_$a -o p -- python $R/bench.py --algo $a --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/bench_${a}_1080p_profiled.json 2> $O/stats_$a.err
  find $O/stats_$a -name "*kernel_stats.csv" -exec cp {} $O/bench_${a}_1080p_kernel_stats.csv \;
  rm -rf $O/stats_$a
  head -4 $O/bench_${a}_1080p_kernel_stats.csv | cut -c1-150
done
python bench.py --algo farn --no-others --no-cpu-baseline > $O/bench_farn_1080p.json 2>> $O/err.log
python bench.py --algo brox --no-others --no-cpu-baseline --no-live-pmc > $O/bench_brox_1080p.json 2>> $O/err.log
python scripts/round5/e2e_stages.py 1920 1080 1537 farn jpg > $O/e2e_stages_farn_jpg.log 2>&1; grep "run\|videos\|stages" $O/e2e_stages_farn_jpg.log
python scripts/round5/e2e_stages.py 1920 1080 1537 tvl1 jpg > $O/e2e_tvl1_one_pipeline.log 2>&1; grep "run 1\|videos" $O/e2e_tvl1_one_pipeline.log | tail -2
if [ -n "$R5_PIPELINES" ]; then # the two-pipeline A/B of the first closing run (profiles/round5/e2e/e2e_tvl1_two_*.log)
python scripts/round5/e2e_stages.py 1920 1080 1537 tvl1 jpg DF_DEVICES=0,0 > $O/e2e_tvl1_two_threads.log 2>&1; grep "run 1\|videos" $O/e2e_tvl1_two_threads.log | tail -2
python scripts/round5/e2e_stages.py 1920 1080 1537 tvl1 jpg DF_DEVICES=0,0 DF_PROCESSES=1 > $O/e2e_tvl1_two_processes.log 2>&1; grep "run 1\|videos" $O/e2e_tvl1_two_processes.log | tail -2
python scripts/round5/e2e_stages.py 1920 1080 1537 farn jpg DF_DEVICES=0,0 DF_PROCESSES=1 > $O/e2e_farn_two_processes.log 2>&1; grep "run 1\|videos" $O/e2e_farn_two_processes.log | tail -2
fi
# PMC traffic of the TVL1 step pair at the bench's batch (the warp kernel changed after scripts/round5/r5_gpu2.sh)
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err
for cnt in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $R/$O/pmc_$cnt -o p -- $R/build/dfx_prof tvl1 1920 1080 /tmp/clip1080.raw 130 1 1 ) > $O/pmc_$cnt.log 2>&1
  python scripts/sq_summary.py $O/pmc_$cnt > $O/pmc_tvl1_$cnt.json 2>&1; rm -rf $O/pmc_$cnt
done
python - <<'PY'
import json
O="gpurun_out/r5_final"
f=json.load(open(O+"/pmc_tvl1_FETCH_SIZE.json")); w=json.load(open(O+"/pmc_tvl1_WRITE_SIZE.json"))
for k in f:
    if "step_fused" in k or "warp" in k: print(k[:50], f[k]["dispatches"], f[k].get("FETCH_SIZE"), w[k].get("WRITE_SIZE"), (2*f[k].get("FETCH_SIZE",0)+w[k].get("WRITE_SIZE",0))*1024/129/1e6, "MB/pair/launch")
PY
python - <<'PY'
import json
O="gpurun_out/r5_final"
d=json.loads(open(O+"/bench_default.json").read().strip().splitlines()[-1])
print("tvl1", d["value"], d["roofline"]["frac"], d["roofline"]["traffic_frac"], d["roofline"]["avg_launch_us"], d.get("parity_check"))
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
for a in ("farn","brox"):
    x=json.loads(open(f"{O}/bench_{a}_1080p.json").read().strip().splitlines()[-1]); print(a, x["value"], x["roofline"]["frac"], x["roofline"]["traffic_frac"], x["roofline"]["avg_launch_us"], x.get("parity_check"), x["config"].get("pcie_png_pairs_per_s"), x["config"].get("pcie_jpeg_pairs_per_s"))
PY
