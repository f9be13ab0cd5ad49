#!/bin/bash
# SQ counters per dispatch for the TVL1 step kernel (one pass, kernel-trace only)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
( SWEEP="0:4:8" timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 9 ) > $R/gpurun_out/pmc_sq.log 2>&1; echo rc=$?
cd $R
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_sq/*counter_collection.csv")
if not f: print("no csv"); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
print(rows[0].keys())
disp=collections.OrderedDict()
for r in rows:
    if "step_fused" not in r["Kernel_Name"]: continue
    d=disp.setdefault(r["Dispatch_Id"],{"grid":r.get("Grid_Size",r.get("Grid_Size_X","")),"start":int(r.get("Start_Timestamp",0)),"end":int(r.get("End_Timestamp",0))})
    d[r["Counter_Name"]]=float(r["Counter_Value"])
items=list(disp.values())
# last 40 dispatches with the largest grid (L0 of the last batch)
mx=max(i["grid"] for i in items)
l0=[i for i in items if i["grid"]==mx][-30:]
names=["SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_INSTS_LDS","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_ACTIVE_INST_VALU"]
print("grid",mx)
for i in l0:
    dur=(i["end"]-i["start"])/1e3
    print(f"dur={dur:7.1f}us "+" ".join(f"{n[3:]}={i.get(n,0):.3g}" for n in names))
PY
rm -rf gpurun_out/pmc_sq
