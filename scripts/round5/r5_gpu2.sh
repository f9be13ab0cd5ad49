#!/bin/bash
# round 5, GPU call 2: the whole GPU suite on the new default arithmetic, the default bench line (parity_check, new legs),
# SQ counters of the step kernel for the three exact hypot readings, PMC traffic at the bench's batch for pmc_traffic.json.
O=gpurun_out/r5_2; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo; R=/root/repo
make -s host > $O/make_host.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_default.json)"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_2/bench_default.json").read().strip().splitlines()[-1])
print("tvl1", d["value"], d["roofline"]["frac"], d["roofline"]["traffic_frac"], d["roofline"]["avg_launch_us"])
print(d.get("parity_check"))
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
for leg in d["config"]["other_workloads"]: print(leg.get("key"), leg.get("pairs_per_s"), leg.get("parity_check"), leg.get("error"), leg.get("wall_s"), leg.get("last_pair_iterations_per_level"), leg.get("step_time_share_per_level"))
print(d.get("cpu_baseline"))
PY
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
run() { n=$1; a=$2; nf=$3; math=$4; shift 4
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof $a 1920 1080 /tmp/clip1080.raw $nf 1 1 0 0 $math ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n > $O/$n.json 2>&1; rm -rf $O/$n; tail -2 $O/$n.log | head -1 | cut -c1-200; }
for m in 0 2 3; do
  run sq_tvl1_math$m tvl1 130 $m SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY
done
for a in tvl1 farn brox; do
  nf=130; [ $a = brox ] && nf=66
  run fetch_$a $a $nf 0 FETCH_SIZE
  run write_$a $a $nf 0 WRITE_SIZE
done
python - <<'PY'
import json
O="gpurun_out/r5_2"
for m in (0,2,3):
    d=json.load(open(f"{O}/sq_tvl1_math{m}.json"))
    for k,v in d.items():
        if "step_fused" in k or "warp" in k:
            print(m, k[:50], {c: (round(x,4) if isinstance(x,float) and x<10 else x) for c,x in v.items() if "/" in c or c in ("dispatches","SQ_INSTS_VALU","SQ_WAVES")})
for a in ("tvl1","farn","brox"):
    f=json.load(open(f"{O}/fetch_{a}.json")); w=json.load(open(f"{O}/write_{a}.json"))
    for k in f:
        fb=f[k].get("FETCH_SIZE",0); wb=w.get(k,{}).get("WRITE_SIZE",0)
        print(a, k[:60], "dispatches", f[k]["dispatches"], "bytes/launch", (2*fb+wb)*1024)
PY
