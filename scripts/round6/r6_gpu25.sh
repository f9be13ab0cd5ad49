#!/bin/bash
# round 6, GPU call 25: SQ / LDS counters of the streaming SOR (early loads in), and the shader clock during it
set -u
R=$GRAFT_REPO_ROOT; O=gpurun_out/r6_25; mkdir -p $R/$O; export TMPDIR=/tmp
cd $R; make -s host > $O/make_host.log 2>&1; make -s build/dfx_prof >> $O/make_host.log 2>&1
python scripts/make_raw_clip.py 1920 1080 2 66 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
run() { n=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof brox 1920 1080 /tmp/clip1080.raw 66 1 1 0 0 0 ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n k_brox_sor_stream > $O/$n.json 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$O/$n/**/*counter_collection.csv", recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "sor_stream" in r["Kernel_Name"]] if f else []
# the longest dispatches = level 0 at 65 pairs
by={}
for r in rows:
    d=by.setdefault(r["Dispatch_Id"],{"dur":(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3})
    d[r["Counter_Name"]]=float(r["Counter_Value"])
top=sorted(by.values(), key=lambda d:-d["dur"])[:10]
if top:
    keys=[k for k in top[0] if k!="dur"]
    print("$n level-0 dispatches (mean of 10): dur_us", round(sum(d["dur"] for d in top)/len(top),1), {k: round(sum(d.get(k,0) for d in top)/len(top)) for k in keys})
PY
  rm -rf $O/$n; }
run sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sqB SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE
run sqC GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
echo "== debug build (cycle counts per phase, workgroups 0 and 77 of the level-0 launches)"
env DFX_LIBRARY=$R/build/variants/libdfx_dbg.so timeout 600 python bench.py --algo brox --frames 66 --steps 1 --warmup 0 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity 2>&1 | grep "sor_stream wg" | head -4
