#!/usr/bin/env python3
"""Per-dispatch timeline of one TVL1 batch from a rocprofv3 --kernel-trace CSV: which launches of which pyramid level
cost what (full K-iteration steps, shorter segment-final steps, warps, launches that find nothing to do).
Usage: python scripts/tvl1_timeline.py <..._kernel_trace.csv> [reduced.csv]  -> markdown on stdout."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if rows and "Start_Timestamp" in rows[0]:  # rocprofv3's own columns
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    k = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          int(r["Start_Timestamp"]) / 1e3, int(r.get("Grid_Size_Z", r.get("Grid_Size", 0)) or 0)) for r in rows]
else:  # the reduced form kept under profiles/ (start_us, dur_us, gx, gy, gz, wg, kernel)
    k = [(r["kernel"], float(r["dur_us"]), float(r["start_us"]), int(r["gz"] or 0)) for r in rows]
k = [x for x in k if x[0].startswith("k_")]
if len(sys.argv) > 2:  # keep the dispatches of this repository's kernels in the reduced form
    with open(sys.argv[2], "w") as f:
        f.write("start_us,dur_us,gx,gy,gz,wg,kernel\n")
        t0 = k[0][2] if k else 0.0
        for name, dur, st, gz in k:
            f.write("%.1f,%.1f,,,%d,,%s\n" % (st - t0, dur, gz, name))
starts = [i for i, x in enumerate(k) if x[0].startswith("k_u8_to_f32")]
last = k[starts[-1]:]
wall = last[-1][2] + last[-1][1] - last[0][2]
pairs = max(x[3] for x in last if x[0].startswith("k_tvl1_step")) or 1
print(f"# TVL1 timeline of the last pass: {pairs} pairs, {wall:.0f} us wall = {wall / pairs:.1f} us per pair "
      f"(sum of kernel durations {sum(x[1] for x in last):.0f} us)\n")
levels = []
for name, dur, st, gz in last:
    if name.startswith("k_tvl1_level_begin"):
        levels.append([])
        continue
    if levels:
        levels[-1].append((name, dur))
tot = {"full": 0.0, "short": 0.0, "warp": 0.0, "noop": 0.0, "other": 0.0}
print("| level (coarsest first) | full steps: n x us | short (segment-final) steps: n x us | warps: n x us | no-op launches: n, us | other us |")
print("|---|---|---|---|---|---|")
for li, L in enumerate(levels):
    steps = [d for n, d in L if n.startswith("k_tvl1_step")]
    warps = [d for n, d in L if n.startswith("k_tvl1_warp")]
    other = sum(d for n, d in L if not (n.startswith("k_tvl1_step") or n.startswith("k_tvl1_warp")))
    big = max(steps) if steps else 0.0
    full = [d for d in steps if d > 0.82 * big]
    short = [d for d in steps if 0.2 * big < d <= 0.82 * big]
    noop_s = [d for d in steps if d <= 0.2 * big]
    wbig = max(warps) if warps else 0.0
    wact = [d for d in warps if d > 0.5 * wbig]
    noop_w = [d for d in warps if d <= 0.5 * wbig]
    tot["full"] += sum(full); tot["short"] += sum(short); tot["warp"] += sum(wact)
    tot["noop"] += sum(noop_s) + sum(noop_w); tot["other"] += other
    f = lambda v: f"{len(v)} x {sum(v) / len(v):.0f}" if v else "0"
    print(f"| {li} | {f(full)} | {f(short)} | {f(wact)} | {len(noop_s) + len(noop_w)}, {sum(noop_s) + sum(noop_w):.0f} | {other:.0f} |")
s = sum(tot.values())
print("\nShare of the batch: " + ", ".join(f"{k_} {100 * v / s:.1f} %" for k_, v in tot.items()) + f" (of {s:.0f} us in level spans)")
