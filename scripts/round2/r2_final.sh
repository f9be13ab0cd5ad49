#!/bin/bash
# Round-2 measurement set: full GPU suite, bench lines (tvl1 / farn / brox 1080p / brox 4K -s=2 / tvl1 224x224), rocprofv3
# kernel-trace stats of the three 1080p bench commands, PMC HBM traffic of the TVL1 step kernel, one bounded attempt at
# the Farneback SQ pass.  Everything lands under gpurun_out/final2/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2
mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -s KILL 600 python bench.py 2>/dev/null | tail -1 > $O/bench_tvl1_1080p.json; echo "tvl1 rc=$?"
timeout -s KILL 400 python bench.py --algo farn 2>/dev/null | tail -1 > $O/bench_farn_1080p.json; echo "farn rc=$?"
timeout -s KILL 400 python bench.py --algo brox --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_brox_1080p.json; echo "brox rc=$?"
timeout -s KILL 400 python bench.py --algo brox --width 3840 --height 2160 --step 2 --frames 34 --steps 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 > $O/bench_brox_4k_s2.json; echo "brox4k rc=$?"
timeout -s KILL 400 python bench.py --width 224 --height 224 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_tvl1_224x224.json; echo "224 rc=$?"
cd /tmp
for a in tvl1 farn brox; do
  ( timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$a -o $a -- python $R/bench.py --algo $a --steps 1 --warmup 1 --frames 100 --no-cpu-baseline --no-pcie ) > $O/rocprof_$a.log 2>&1; echo "rocprof $a rc=$?"
  F=$(find $O/prof_$a -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && grep -E "Name|k_tvl1|k_farn|k_brox|k_u8|k_pyr|k_cent|k_flow" "$F" > $O/bench_${a}_1080p_kernel_stats.csv
  tail -1 $O/rocprof_$a.log | cut -c1-3000 > $O/bench_${a}_under_rocprof.json
  rm -rf $O/prof_$a
done
cd $R
ALGOS="tvl1" bash scripts/gpu_pmc.sh > $O/gpu_pmc.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_tvl1.json; tail -14 $O/gpu_pmc.log | cut -c1-160
cd /tmp
( export ALGO=farn SWEEP=0:0:8:0; timeout -s KILL 100 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq_farn_A -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 9 ) > $O/sq_farn_A.log 2>&1; echo "sq farn rc=$?"
python $R/scripts/sq_summary.py $O/sq_farn_A k_farn_iteration > $O/sq_farn_A.json 2>>$O/sq_farn_A.log; rm -rf $O/sq_farn_A; head -20 $O/sq_farn_A.json
cd $R
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; c=d.get("cpu_baseline") or {}; p=d.get("pcie_inclusive") or {}
    print(d["metric"], round(d["value"],1), r["kernel"], "frac", round(r["frac"],3), "traffic_frac", r.get("traffic_frac"), "pcie", p.get("value"), "cpu", c.get("value"), c.get("cores"))
except Exception as e:
    print("unparsed", e)
PY
done
for a in tvl1 farn brox; do echo "== kernel stats $a"; cut -c1-150 $O/bench_${a}_1080p_kernel_stats.csv | head -12; done
