#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of a PCIe-inclusive FlowBuffer pass: per direction the
number of copies, their durations, the busy time of the copy engine and the gaps between copies, next to the kernels'
busy time — is the host link the limit, or the way the copies are issued?
    python scripts/copy_timeline.py <dir with *_memory_copy_trace.csv and *_kernel_trace.csv> [t_from_ms t_to_ms]"""
import csv
import glob
import sys

import numpy as np

d = sys.argv[1]
cp = list(csv.DictReader(open(glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True)[0])))
kt = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0])))
t0 = min(int(r["Start_Timestamp"]) for r in cp + kt)
ks = np.array([(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in kt], dtype=np.float64) * 1e-6
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1e18)


def union(iv):
    iv = iv[np.argsort(iv[:, 0])]
    tot, cs, ce = 0.0, iv[0, 0], iv[0, 1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


print(f"kernels: {len(ks)} dispatches, span {ks[:,0].min():.1f} .. {ks[:,1].max():.1f} ms, busy {union(ks):.1f} ms")
for direction in ("MEMORY_COPY_HOST_TO_DEVICE", "MEMORY_COPY_DEVICE_TO_HOST", "MEMORY_COPY_DEVICE_TO_DEVICE"):
    iv = np.array([(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in cp if r["Direction"] == direction],
                  dtype=np.float64) * 1e-6
    if not len(iv):
        continue
    iv = iv[(iv[:, 0] >= lo) & (iv[:, 1] <= hi)]
    dur = iv[:, 1] - iv[:, 0]
    srt = iv[np.argsort(iv[:, 0])]
    gaps = srt[1:, 0] - srt[:-1, 1]
    print(f"{direction}: {len(iv)} copies, span {iv[:,0].min():.1f} .. {iv[:,1].max():.1f} ms, busy {union(iv):.1f} ms; "
          f"duration median {np.median(dur)*1e3:.0f} us (p10 {np.percentile(dur,10)*1e3:.0f}, p90 {np.percentile(dur,90)*1e3:.0f}, "
          f"max {dur.max()*1e3:.0f}); gap between consecutive copies median {np.median(gaps)*1e3:.0f} us, "
          f"p90 {np.percentile(gaps,90)*1e3:.0f} us, gaps > 1 ms: {(gaps > 1.0).sum()}")
    # bursts: runs of copies separated by < 1 ms
    b0 = 0
    for i in range(len(gaps) + 1):
        if i == len(gaps) or gaps[i] > 1.0:
            seg = srt[b0:i + 1]
            print(f"    burst of {len(seg):4d} copies {seg[0,0]:8.1f} .. {seg[-1,1]:8.1f} ms ({seg[-1,1]-seg[0,0]:6.1f} ms, "
                  f"{(seg[:,1]-seg[:,0]).sum():6.1f} ms inside copies)")
            b0 = i + 1
