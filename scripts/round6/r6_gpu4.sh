#!/bin/bash
# round 6, GPU call 4: counters of the head kernel (3-plane image tile, p loaded after the warp) through tools/dfx_prof
set -u
R=$GRAFT_REPO_ROOT; O=gpurun_out/r6_4; mkdir -p $R/$O; export TMPDIR=/tmp
cd $R; make -s host > $O/make_host.log 2>&1 || tail -5 $O/make_host.log
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
run() { n=$1; var=$2; shift 2
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof tvl1 1920 1080 /tmp/clip1080.raw 130 1 1 0 $var 0 ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n > $O/$n.json 2>&1; rm -rf $O/$n; tail -2 $O/$n.log | head -1 | cut -c1-160; }
run sqA_head 0 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sqB_head 0 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch_head 0 FETCH_SIZE
run write_head 0 WRITE_SIZE
run sqA_nohead 64 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run fetch_nohead 64 FETCH_SIZE
run write_nohead 64 WRITE_SIZE
python - <<'PY'
import json
O="gpurun_out/r6_4"
for n in ("sqA_head","sqB_head","sqA_nohead"):
    d=json.load(open(f"{O}/{n}.json"))
    for k,v in d.items():
        if "tvl1_step" in k or "warp" in k:
            print(n, k[:40], {c: (round(x,4) if isinstance(x,float) and x<10 else round(x)) for c,x in v.items() if c!="note"})
for n in ("head","nohead"):
    f=json.load(open(f"{O}/fetch_{n}.json")); w=json.load(open(f"{O}/write_{n}.json"))
    for k in f:
        if "tvl1_step" in k or "warp" in k:
            fb=f[k].get("FETCH_SIZE",0); wb=w.get(k,{}).get("WRITE_SIZE",0)
            print(n, k[:40], "dispatches", f[k]["dispatches"], "MB/launch fetch(x2 corrected)", round(2*fb*1024/1e6,1), "write", round(wb*1024/1e6,1))
PY
