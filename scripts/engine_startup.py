#!/usr/bin/env python3
"""Time dfx_create + the first (tiny) FlowBuffer for an engine size and batch capacity: what the automatic batch costs at
start-up.  usage: engine_startup.py W H algo max_batch [max_batch ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DFX_NO_TORCH", "1")
import denseflow_amd as dfx  # noqa: E402
from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, algo = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
frames = SynthClip(W, H, 1).frames(3)
with dfx.FlowEngine(W, H, algo, max_batch=1) as e:  # the process's own HIP start-up, once
    e.calc_optflows_u8(frames, 1, 20)
for mb in [int(x) for x in sys.argv[4:]]:
    t0 = time.perf_counter()
    e = dfx.FlowEngine(W, H, algo, max_batch=mb)
    t1 = time.perf_counter()
    e.calc_optflows_jpeg(frames, 1, 20)
    t2 = time.perf_counter()
    e.calc_optflows_jpeg(frames, 1, 20)
    t3 = time.perf_counter()
    e.close()
    print(f"{algo} {W}x{H} max_batch {mb}: create {1e3 * (t1 - t0):.1f} ms, first FlowBuffer {1e3 * (t2 - t1):.1f} ms, "
          f"second {1e3 * (t3 - t2):.1f} ms, close {1e3 * (time.perf_counter() - t3):.1f} ms")
