#!/bin/bash
# round 6, GPU call 23: streaming SOR — the next tile's u / v / du / dv loads issued 1, 2, 3 sweeps before the end of the tile
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_23; mkdir -p $O; export TMPDIR=/tmp; cd $R
B="--steps 3 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc"
for rep in 1 2; do
for e in 0 1 2 3; do
  L=""; [ $e != 0 ] && L="DFX_LIBRARY=$R/build/variants/libdfx_e$e.so"
  env $L timeout 600 python bench.py --algo brox --frames 131 $B 2> $O/err_$e.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1080p early $e:', d['value'], d.get('parity_check',{}).get('max_abs'))"
done; done
# per-launch durations of the streaming kernel (which launches are the level-0 ones, and how long they take)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python $R/bench.py --algo brox --frames 131 --steps 1 --warmup 0 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/profiled.json 2> $O/trace.err
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \; ; rm -rf $O/trace
python - <<'PY'
import csv,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6_23'
rows=[r for r in csv.DictReader(open(O+'/kernel_trace.csv')) if 'sor_stream' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
h=collections.Counter(round(x,-2) for x in d)
print(len(d),'launches; histogram of durations (us, rounded to 100):',sorted(h.items()))
PY
head -c 20000000 $O/kernel_trace.csv > /dev/null; rm -f $O/kernel_trace.csv
