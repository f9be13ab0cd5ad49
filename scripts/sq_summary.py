#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch and the
derived busy fractions.  Usage: sq_summary.py <dir-or-csv> [kernel-substring ...] ; prints JSON."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src = sys.argv[1]
    pats = sys.argv[2:] or [""]
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    grid = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            full = r["Kernel_Name"]
            if not any(p in full for p in pats):
                continue
            k = full.replace("(anonymous namespace)::", "").split("(")[0]
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            grid[k] = max(grid.get(k, 0), int(r.get("Grid_Size", 0) or 0))
    out = {}
    for k, cs in agg.items():
        d = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
        d["dispatches"] = max(v[0] for v in cs.values())
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS",
                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU"):
                if c in d:
                    d[c + "/WAVE_CYCLES"] = d[c] / wc
        if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
            d["note"] = "SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md)"
        out[k] = d
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
