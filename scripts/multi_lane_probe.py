#!/usr/bin/env python3
"""Do HBM-bound launches (warps, 2-iteration steps) of one batch overlap VALU-bound launches (full steps) of another
when two or three handles work on one GPU from separate host threads (private streams)?
Usage: python scripts/multi_lane_probe.py [W H NF]; ALGO=tvl1|farn|brox; LANES="1,2,3"; prints one line per lane count."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import denseflow_amd  # noqa: E402
from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, NF = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1920, 1080, 300)))
ALGO = os.environ.get("ALGO", "tvl1")
LANES = [int(v) for v in os.environ.get("LANES", "1,2,3,1,2").split(",")]
REPS = int(os.environ.get("REPS", "2"))
dev = torch.device("cuda", 0)
clip = SynthClip(W, H, 2)
d_frames = clip.frames_torch(NF, dev)
d_flows = torch.empty((NF - 1, H, W, 2), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ref = None
M = NF - 1
for lanes in LANES:
    # lane l computes flows [lo, hi): frames [lo, hi]
    cuts = [M * l // lanes for l in range(lanes + 1)]
    per = max(cuts[l + 1] - cuts[l] for l in range(lanes))
    mb = int(os.environ.get("MAXB", "0")) or min(per, 129)
    engs = [denseflow_amd.FlowEngine(W, H, ALGO, max_batch=mb) for _ in range(lanes)]

    def work(l, reps):
        lo, hi = cuts[l], cuts[l + 1]
        for _ in range(reps):
            engs[l].calc_optflows_device(d_frames.data_ptr() + lo * W * H, W, W * H, hi - lo + 1, 1,
                                         d_flows.data_ptr() + lo * W * H * 8, W * H * 2)

    def run(reps):
        ts = [threading.Thread(target=work, args=(l, reps)) for l in range(lanes)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    d_flows.zero_()
    run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(REPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = d_flows.clone()
    same = "ref" if ref is None else ("bit-identical" if torch.equal(out, ref) else "DIFFERENT")
    if ref is None:
        ref = out
    print(f"{ALGO} {W}x{H} lanes={lanes} max_batch={mb}: {REPS * M / dt:8.1f} pairs/s  [{same}]", flush=True)
    for e in engs:
        e.close()
