#!/bin/bash
# HBM traffic of the dominant kernels from the PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: calibrated on 4 B/lane and 16 B/lane streaming copies of known size,
# profiles/round2/pmc/README.md),
# kernel-trace only (MI355X_MICROARCH.md §HBM).  PMC_BATCH pairs per launch (default 16 with 34 frames: rocprofv3 --pmc was
# unstable on longer runs on this pool in round 2; round 3 measures at the bench's own batch, PMC_BATCH=129, one batch).
# Writes gpurun_out/pmc_traffic.json in the layout of profiles/pmc_traffic.json.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ALGOS=${ALGOS:-"tvl1 farn"}
export PMC_BATCH=${PMC_BATCH:-16}
PMC_FRAMES=$(( PMC_BATCH == 16 ? 34 : PMC_BATCH + 1 ))
cd /tmp
for A in $ALGOS; do
for CNT in FETCH_SIZE WRITE_SIZE; do
  ( timeout -s KILL ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_${A}_$CNT -o p -- python $R/bench.py --algo $A --steps 1 --warmup 0 --frames $PMC_FRAMES --max-batch $PMC_BATCH --no-cpu-baseline --no-others --no-pcie ) > $R/gpurun_out/pmc_${A}_$CNT.log 2>&1; echo "pmc $A $CNT rc=$?"
done; done
cd $R
ALGOS="$ALGOS" python - <<'PY'
import csv,glob,collections,json,os
dom={"tvl1":"void k_tvl1_step_fused<true, 0>","farn":"void k_farn_iteration_t<6>","brox":"void k_brox_sor_pk<5>"}
B=int(os.environ.get("PMC_BATCH","16"))
comp={"tvl1":"void k_tvl1_warp<5>"}  # launched in front of every step launch: one pair of launches per step
out={}
for A in os.environ["ALGOS"].split():
    per={}
    for CNT in ("FETCH_SIZE","WRITE_SIZE"):
        fs=glob.glob(f"gpurun_out/pmc_{A}_{CNT}/**/*counter_collection.csv", recursive=True)
        if not fs: print("no csv",A,CNT); continue
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(fs[0])):
            if r.get("Counter_Name")!=CNT: continue
            k=r["Kernel_Name"].split("(")[0]
            agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
        for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:5]:
            print(A,CNT,k[:50],v)
        per[CNT]=agg.get(dom[A])
        if A in comp: per["c_"+CNT]=agg.get(comp[A])
    if per.get("FETCH_SIZE") and per.get("WRITE_SIZE"):
        n=per["FETCH_SIZE"][0]; f=per["FETCH_SIZE"][1]/n; w=per["WRITE_SIZE"][1]/per["WRITE_SIZE"][0]
        b=(2*f+w)*1024
        out[A]={"kernel":dom[A],"launches":n,"FETCH_SIZE_KB_per_launch":f,"WRITE_SIZE_KB_per_launch":w,
                "hbm_bytes_per_launch":b,"measured_batch":B,"hbm_bytes_per_launch_per_pair":b/B,
                "workload":("1920x1080, 34 frames (33 pairs, batches of 16/16/1), 1 GPU" if B==16 else f"1920x1080, {B+1} frames = one batch of {B} pairs (the batch bench.py runs), 1 GPU")}
        if per.get("c_FETCH_SIZE") and per.get("c_WRITE_SIZE"):
            cf=per["c_FETCH_SIZE"][1]/per["c_FETCH_SIZE"][0]; cw=per["c_WRITE_SIZE"][1]/per["c_WRITE_SIZE"][0]
            out[A].update({"companion_kernel":comp[A],"companion_launches":per["c_FETCH_SIZE"][0],
                           "companion_FETCH_SIZE_KB_per_launch":cf,"companion_WRITE_SIZE_KB_per_launch":cw,
                           "companion_hbm_bytes_per_launch_per_pair":(2*cf+cw)*1024/B})
json.dump(out,open("gpurun_out/pmc_traffic.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
rm -rf gpurun_out/pmc_*_FETCH_SIZE gpurun_out/pmc_*_WRITE_SIZE
