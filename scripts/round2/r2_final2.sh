#!/bin/bash
# Round-2 final measurement set (after the tile-geometry / XCD-mapping / Farneback + Brox changes): full GPU suite, smoke,
# PMC HBM traffic of the three dominant kernels (merged into profiles/pmc_traffic.json BEFORE the bench lines so that
# `roofline.traffic` is the traffic of the kernels that are timed), the bench lines, rocprofv3 kernel stats of the three
# 1080p bench commands, the per-dispatch timeline of one TVL1 batch and the SQ counters of the step kernel.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
PMC_TIMEOUT=150 ALGOS="tvl1 farn brox" bash scripts/gpu_pmc.sh > $O/gpu_pmc.log 2>&1; tail -3 $O/gpu_pmc.log | cut -c1-200
python - <<'PY'
import json
new = json.load(open("gpurun_out/pmc_traffic.json"))
old = json.load(open("profiles/pmc_traffic.json"))
for a, d in new.items():
    keep = {k: old.get(a, {}).get(k) for k in ("how", "note") if old.get(a, {}).get(k)}
    old[a] = dict(d, **keep)
    old[a]["measured"] = "scripts/r2_final2.sh (final kernels of round 2)"
json.dump(old, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(old, open("gpurun_out/final3/pmc_traffic.json", "w"), indent=1)
print({a: round(d["hbm_bytes_per_launch_per_pair"] / 1e6, 2) for a, d in old.items()}, "MB per pair and launch;", sorted(new))
PY
timeout -s KILL 600 python bench.py 2>/dev/null | tail -1 > $O/bench_tvl1_1080p.json; echo "tvl1 rc=$?"
timeout -s KILL 400 python bench.py --algo farn 2>/dev/null | tail -1 > $O/bench_farn_1080p.json; echo "farn rc=$?"
timeout -s KILL 400 python bench.py --algo brox --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_brox_1080p.json; echo "brox rc=$?"
timeout -s KILL 400 python bench.py --algo brox --width 3840 --height 2160 --step 2 --frames 34 --steps 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 > $O/bench_brox_4k_s2.json; echo "brox4k rc=$?"
timeout -s KILL 400 python bench.py --width 224 --height 224 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_tvl1_224x224.json; echo "224 rc=$?"
cd /tmp
for a in tvl1 farn brox; do
  ( timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$a -o $a -- python $R/bench.py --algo $a --steps 1 --warmup 1 --frames 100 --no-cpu-baseline --no-pcie ) > $O/rocprof_$a.log 2>&1; echo "rocprof $a rc=$?"
  F=$(find $O/prof_$a -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && grep -E "Name|k_tvl1|k_farn|k_brox|k_u8|k_pyr|k_cent|k_flow" "$F" > $O/bench_${a}_1080p_kernel_stats.csv
  tail -1 $O/rocprof_$a.log | cut -c1-3000 > $O/bench_${a}_under_rocprof.json
  rm -rf $O/prof_$a
done
# per-dispatch timeline of one 129-pair TVL1 batch
( SWEEP="0:4:0:0" timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/scripts/sweep_tvl1.py 1920 1080 130 ) > $O/trace.log 2>&1; echo "trace rc=$?"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && python $R/scripts/tvl1_timeline.py "$F" $O/tvl1_timeline_dispatches.csv > $O/tvl1_timeline.md 2>$O/tvl1_timeline.err; rm -rf $O/trace; head -30 $O/tvl1_timeline.md
# SQ counters of the step kernel, real schedule (batches of 16, as in profiles/round2/tvl1_step/README.md)
for P in "A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  T=${P%% *}; C=${P#* }
  ( SWEEP="0:4:16:0" timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq_$T -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 17 ) > $O/sq_$T.log 2>&1; echo "sq $T rc=$?"
  python $R/scripts/sq_summary.py $O/sq_$T step_fused > $O/sq_final_kernel_$T.json 2>>$O/sq_$T.log; rm -rf $O/sq_$T
done
( ALGO=brox SWEEP="0:0:8:0" timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq_brox -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 9 ) > $O/sq_brox.log 2>&1; echo "sq brox rc=$?"
python $R/scripts/sq_summary.py $O/sq_brox k_brox_sor_fused > $O/sq_brox_sor_final_A.json 2>>$O/sq_brox.log; rm -rf $O/sq_brox
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
