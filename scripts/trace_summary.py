#!/usr/bin/env python3
"""Compress a rocprofv3 kernel trace CSV into per-(kernel, grid) duration lists for the last batch."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_u8_to_f32"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
seq = rows[idx[-1]:]
t0 = int(seq[0]["Start_Timestamp"])
cur, line, end = None, [], 0
for r in seq:
    n = r["Kernel_Name"]
    if "at::" in n or "rocclr" in n:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    end = max(end, (int(r["End_Timestamp"]) - t0) / 1e3)
    key = (n.split("(")[0][:26], r["Grid_Size_X"])
    if key != cur:
        if line:
            print(cur[0], cur[1], " ".join(f"{x:.0f}" for x in line))
        cur, line = key, []
    line.append(d)
print(cur[0], cur[1], " ".join(f"{x:.0f}" for x in line))
print("batch total ms", end / 1e3)
