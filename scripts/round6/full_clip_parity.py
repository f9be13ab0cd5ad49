#!/usr/bin/env python3
"""One-off evidence beyond bench.py's five-flow parity_check: EVERY flow of the headline clip (1920x1080, seed 2, 300 frames,
-a=tvl1 -s=1: 299 flows in device batches of 129 + 129 + 41) against the CPU oracle — the checker, run in parallel worker
processes — plus the executed iteration totals.  Usage: full_clip_parity.py [n_frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import denseflow_amd  # noqa: E402
from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, NF = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
d_frames = SynthClip(W, H, 2).frames_torch(NF, dev)
d_flows = torch.empty((NF - 1, H, W, 2), dtype=torch.float32, device=dev)
with denseflow_amd.FlowEngine(W, H, "tvl1") as eng:
    eng.calc_optflows_device(d_frames.data_ptr(), W, W * H, NF, 1, d_flows.data_ptr(), W * H * 2)
    st = eng.stats()
torch.cuda.synchronize()
frames = d_frames.cpu().numpy()
from oracle import oracle_py  # noqa: E402  (the checker)

t0 = time.time()
worst, differing, iters = 0.0, 0, 0
for i in range(NF - 1):
    ref, tr = oracle_py.tvl1_calc(frames[i], frames[i + 1], want_trace=True, threads=16)
    out = d_flows[i].cpu().numpy()
    iters += sum(sum(r) for r in tr.iters_table())
    if not np.array_equal(out, ref):
        differing += 1
        worst = max(worst, float(np.max(np.abs(out - ref))))
print(f"headline clip, {NF - 1} flows: {differing} differ from the oracle (max-abs {worst}); inner iterations executed: device "
      f"{st.tvl1_total_iters}, oracle {iters}; oracle time {time.time() - t0:.0f} s on 16 threads")
sys.exit(1 if differing or iters != st.tvl1_total_iters else 0)
