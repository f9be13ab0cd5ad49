#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
echo "== farn batch sweep"; ALGO=farn SWEEP_LEVELS=1 SWEEP="0:0:1,0:0:2,0:0:16" timeout -s KILL 200 python scripts/sweep_tvl1.py 2>&1 | grep -v amdgpu
echo "== tvl1 224x224 (config 4 shape)"; (timeout -s KILL 200 python bench.py --width 224 --height 224 --frames 300 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1) | tee gpurun_out/bench_tvl1_224.log | cut -c1-900
echo "== brox trace 1080p"; cd /tmp; ( ALGO=brox SWEEP="0:0:8" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_brox -o b -- python $R/scripts/sweep_tvl1.py 1920 1080 9 ) > $R/gpurun_out/rocprof_brox.log 2>&1; cd $R
grep -E "k_brox|Name" gpurun_out/prof_brox/b_kernel_stats.csv | cut -c1-150
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/prof_brox/b_kernel_trace.csv")))
idx=[i for i,r in enumerate(rows) if "k_brox_u8_to_f32" in r["Kernel_Name"]]
seq=[r for r in rows[idx[-1]:] if "k_brox" in r["Kernel_Name"]]
# per-level time: levels identified by grid size of k_brox_stage1
lv=collections.OrderedDict()
cur=None
for r in seq:
    n=r["Kernel_Name"].split("(")[0]; g=r["Grid_Size_X"]+"x"+r["Grid_Size_Y"]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if n=="k_brox_level_init": cur=g; lv[cur]=collections.Counter()
    if cur: lv[cur][n]+=d
t0=int(seq[0]["Start_Timestamp"]); t1=int(seq[-1]["End_Timestamp"])
print("batch wall ms",(t1-t0)/1e6)
for g,c in lv.items(): print(g, " ".join(f"{k[7:]}={v:.0f}us" for k,v in c.items()), "sum=%.0f"%sum(c.values()))
PY
rm -f gpurun_out/prof_brox/*trace.csv
