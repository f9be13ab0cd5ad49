#!/bin/bash
# Round 4: HBM traffic of the dominant kernels AT THE BENCH'S OWN BATCH, torch-free, and the
# blocking-sync A/B (item 7).  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes around tools/dfx_prof.
O=gpurun_out/r4_pmc; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo; R=/root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
# --- blocking-sync A/B: pairs/s and CPU-ms per pair (user + system, all threads) ---
for a in tvl1 farn brox; do for bs in 0 1 0 1; do
  n=130; [ $a = brox ] && n=66
  ./build/dfx_prof $a 1920 1080 /tmp/clip1080.raw $n 1 2 0 0 0 $bs >> $O/blocking_sync_ab.txt 2>> $O/err.log
done; done
grep -o '"algo":"[a-z0-9]*"\|"pairs_per_s":[0-9.]*\|"cpu_ms_per_pair":[0-9.]*\|"cpu_busy_fraction":[0-9.]*\|"blocking_sync":[01]' $O/blocking_sync_ab.txt | paste - - - - -
# --- PMC traffic ---
run() { n=$1; a=$2; nf=$3; shift 3
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof $a 1920 1080 /tmp/clip1080.raw $nf 1 1 ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n > $O/$n.json 2>&1; rm -rf $O/$n; tail -2 $O/$n.log | head -1; }
for a in tvl1 farn brox; do
  nf=130; [ $a = brox ] && nf=66
  run fetch_$a $a $nf FETCH_SIZE
  run write_$a $a $nf WRITE_SIZE
done
python - <<'PY'
import json,glob
O="gpurun_out/r4_pmc"
for a in ("tvl1","farn","brox"):
    f=json.load(open(f"{O}/fetch_{a}.json")); w=json.load(open(f"{O}/write_{a}.json"))
    for k in f:
        fb=f[k].get("FETCH_SIZE",0); wb=w.get(k,{}).get("WRITE_SIZE",0)
        print(a, k[:60], "dispatches", f[k]["dispatches"], "bytes/launch", (2*fb+wb)*1024)
PY
