#!/usr/bin/env python3
"""One 1080p Farneback FlowBuffer (130 frames) through dfx_calc_batch_jpeg a few times: the rate with complete JPEG files
out, and — run under `rocprofv3 --kernel-trace --stats` — the cost of the five JPEG launches beside the flow kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import denseflow_amd as dfx  # noqa: E402
from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, N = 1920, 1080, 130
frames = SynthClip(W, H, 2).frames(N)
with dfx.FlowEngine(W, H, "farn") as eng:
    eng.calc_optflows_jpeg(frames, 1, 20, 95)
    t = time.perf_counter()
    for _ in range(3):
        jx, jy = eng.calc_optflows_jpeg(frames, 1, 20, 95)
    dt = (time.perf_counter() - t) / 3
    print(f"farn 1080p, JPEG files out (host frames in): {(N - 1) / dt:.1f} pairs/s, mean file {sum(map(len, jx + jy)) / (2 * (N - 1)):.0f} bytes")
