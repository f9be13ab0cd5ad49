#!/bin/bash
# Round-end measurement set: bench lines (tvl1 / farn / brox / tvl1 224x224) and rocprofv3 kernel-trace stats
# of the tvl1 and farn bench commands.  Everything lands under gpurun_out/final/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -s KILL 600 python bench.py 2>/dev/null | tail -1 > $O/bench_tvl1_1080p.json; echo "tvl1 rc=$?"
timeout -s KILL 400 python bench.py --algo farn 2>/dev/null | tail -1 > $O/bench_farn_1080p.json; echo "farn rc=$?"
timeout -s KILL 400 python bench.py --algo brox --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_brox_1080p.json; echo "brox rc=$?"
timeout -s KILL 400 python bench.py --width 224 --height 224 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_tvl1_224x224.json; echo "224 rc=$?"
cd /tmp
for a in tvl1 farn brox; do
  ( timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$a -o $a -- python $R/bench.py --algo $a --steps 1 --warmup 1 --frames 100 --no-cpu-baseline ) > $O/rocprof_$a.log 2>&1; echo "rocprof $a rc=$?"
  F=$(find $O/prof_$a -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && grep -E "Name|k_tvl1|k_farn|k_brox|k_u8|k_pyr|k_cent|k_flow" "$F" > $O/bench_${a}_1080p_kernel_stats.csv
  tail -1 $O/rocprof_$a.log | cut -c1-2000 > $O/bench_${a}_under_rocprof.json
  rm -rf $O/prof_$a
done
cd $R
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["metric"], round(d["value"],1), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("unparsed", e)
PY
done
for a in tvl1 farn brox; do echo "== kernel stats $a"; cut -c1-140 $O/bench_${a}_1080p_kernel_stats.csv | head -14; done
