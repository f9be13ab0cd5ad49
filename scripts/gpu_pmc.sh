#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, kernel-trace only) for the bench workload
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for A in tvl1 farn; do
for CNT in FETCH_SIZE WRITE_SIZE; do
  ( timeout 600 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_${A}_$CNT -o p -- python $R/bench.py --algo $A --steps 1 --warmup 0 --frames 66 --no-cpu-baseline ) > $R/gpurun_out/pmc_${A}_$CNT.log 2>&1; echo "pmc $A $CNT rc=$?"
done; done
cd $R
python - <<'PY'
import csv,glob,collections,json
out={}
for A in ("tvl1","farn"):
    res={}
    for CNT in ("FETCH_SIZE","WRITE_SIZE"):
        fs=glob.glob(f"gpurun_out/pmc_{A}_{CNT}/*counter_collection.csv")
        if not fs: print("no csv",A,CNT); continue
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(fs[0])):
            if r.get("Counter_Name")!=CNT: continue
            k=r["Kernel_Name"].split("(")[0]
            agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
        res[CNT]={k:{"launches":v[0],"sum":v[1],"per_launch":v[1]/max(v[0],1)} for k,v in agg.items() if k.startswith(("k_","void k_"))}
    out[A]=res
json.dump(out,open("gpurun_out/pmc_summary.json","w"),indent=1)
for A,res in out.items():
    for CNT,d in res.items():
        for k,v in sorted(d.items(), key=lambda kv:-kv[1]["sum"])[:4]:
            print(A,CNT,k[:40],v)
PY
rm -rf gpurun_out/pmc_*_FETCH_SIZE gpurun_out/pmc_*_WRITE_SIZE
