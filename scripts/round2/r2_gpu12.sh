#!/bin/bash
# round 2, GPU call 12: per-dispatch timeline of a 129-pair TVL1 batch at 1080p (which launches of which level cost what)
mkdir -p gpurun_out/r2l; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l
cd /tmp
for G in 3 0; do
  ( SWEEP="0:4:0:0:$G" timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_g$G -o t -- python $R/scripts/sweep_tvl1.py 1920 1080 130 ) > $O/trace_g$G.log 2>&1; echo "trace geom=$G rc=$?"
  F=$(find $O/trace_g$G -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then
    python - "$F" $O/trace_g$G.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,dur_us,gx,gy,gz,wg,kernel\n")
    for r in rows:
        n = r["Kernel_Name"]
        n = n.split("(")[0].replace("void ", "")[:48]
        f.write("%.1f,%.1f,%s,%s,%s,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), n))
print(len(rows), "dispatches")
PY
  fi
  rm -rf $O/trace_g$G
  grep -v amdgpu.ids $O/trace_g$G.log | tail -2 | cut -c1-200
done
ls -la $O
