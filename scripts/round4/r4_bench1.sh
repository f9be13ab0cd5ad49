#!/bin/bash
# Round 4: blocking-sync A/B again (device flag), then the whole default bench line (live PMC legs included)
O=gpurun_out/r4_bench1; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for a in tvl1 farn; do for bs in 0 1 0 1; do
  ./build/dfx_prof $a 1920 1080 /tmp/clip1080.raw 130 1 2 0 0 0 $bs >> $O/blocking_sync_ab.txt 2>> $O/err.log
done; done
grep -o '"algo":"[a-z0-9]*"\|"pairs_per_s":[0-9.]*\|"cpu_ms_per_pair":[0-9.]*\|"cpu_busy_fraction":[0-9.]*\|"blocking_sync":[01]' $O/blocking_sync_ab.txt | paste - - - - -
/usr/bin/time -v python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
grep -E "Elapsed|Maximum resident" $O/bench_default.err
wc -c $O/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench1/bench_default.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:900])
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
print(d["cpu_baseline"])
PY
