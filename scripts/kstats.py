#!/usr/bin/env python3
"""Print this repository's kernels from a rocprofv3 *kernel_stats.csv: name, calls, total ms, average us."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_" in r["Name"] and "at::" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t = float(r["TotalDurationNs"])
    name = r["Name"].replace("(anonymous namespace)::", "")
    print(f'{name.split("(")[0][:44]:44s} calls {int(r["Calls"]):6d}  total {t / 1e6:9.3f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {100 * t / tot:5.1f} %')
print(f'{"sum":44s} {"":12s}  total {tot / 1e6:9.3f} ms')
