#!/usr/bin/env python3
"""Per-batch compute windows and copy-kernel statistics from a rocprofv3 --kernel-trace of scripts/pcie_path_probe.py.
    python scripts/pcie_windows.py <trace dir> <flow-kernel name prefix, e.g. k_farn>"""
import csv
import glob
import sys

import numpy as np

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0])))
pre = sys.argv[2]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
iv = lambda f: np.array([(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in rows if f(r["Kernel_Name"])],
                        dtype=np.float64) * 1e-6
comp, cp = iv(lambda n: pre in n), iv(lambda n: "copyBuffer" in n or "k_egress" in n)
comp = comp[np.argsort(comp[:, 0])]
g = comp[1:, 0] - comp[:-1, 1]
b0, out = 0, []
for i in range(len(g) + 1):
    if i == len(g) or g[i] > 3.0:
        seg = comp[b0:i + 1]
        out.append((seg[0, 0], seg[-1, 1], len(seg)))
        b0 = i + 1
print("compute windows (ms): " + "  ".join(f"[{a:.0f}..{b:.0f} = {b-a:.1f}, {n} k]" for a, b, n in out[-12:]))
if len(cp):
    d = cp[:, 1] - cp[:, 0]
    big = d > 0.05
    print(f"copy kernels: {len(cp)} ({big.sum()} over 50 us); long ones: total {d[big].sum():.1f} ms, median {np.median(d[big])*1e3:.0f} us, "
          f"max {d[big].max()*1e3:.0f} us")
    for a, b in cp[big][-8:]:
        print(f"    copy kernel {a:.1f}..{b:.1f} ms ({b-a:.2f} ms)")
