#!/usr/bin/env python3
"""One FlowBuffer through the host-pointer entry points with page-locked buffers, float flows out — the PCIe-inclusive
path on its own, for timelines (rocprofv3 --kernel-trace --memory-copy-trace) and A/B runs.
    python scripts/pcie_path_probe.py ALGO VARIANT EGRESS_WGS [W H NF PASSES] ; prints pairs/s of every pass."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import denseflow_amd  # noqa: E402
from denseflow_amd.synth import SynthClip  # noqa: E402

algo, variant, wgs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
W, H, NF, PASSES = (int(v) for v in (sys.argv[4:8] if len(sys.argv) >= 8 else (1920, 1080, 300, 3)))
u8 = os.environ.get("OUT") == "u8"
L = denseflow_amd.load_library()
dev = torch.device("cuda", 0)
d_frames = SynthClip(W, H, 2).frames_torch(NF, dev)
h_frames = torch.empty((NF, H, W), dtype=torch.uint8, pin_memory=True)
h_frames.copy_(d_frames)
M = NF - 1
fp = (C.c_void_p * NF)(*[h_frames[i].data_ptr() for i in range(NF)])
if u8:
    h_x = torch.empty((M, H, W), dtype=torch.uint8, pin_memory=True)
    h_y = torch.empty((M, H, W), dtype=torch.uint8, pin_memory=True)
    xp = (C.c_void_p * M)(*[h_x[i].data_ptr() for i in range(M)])
    yp = (C.c_void_p * M)(*[h_y[i].data_ptr() for i in range(M)])
else:
    h_flows = torch.empty((M, H, W, 2), dtype=torch.float32, pin_memory=True)
    op = (C.c_void_p * M)(*[h_flows[i].data_ptr() for i in range(M)])
torch.cuda.synchronize()
eng = denseflow_amd.FlowEngine(W, H, algo, variant=variant, egress_workgroups=wgs)
for p in range(PASSES):
    t0 = time.perf_counter()
    if u8:
        rc = L.dfx_calc_batch_u8(eng._h, fp, W, NF, 1, -20.0, 20.0, xp, yp, W)
    else:
        rc = L.dfx_calc_batch(eng._h, fp, W, NF, 1, op, W * 8)
    assert rc == 0, L.dfx_last_error(eng._h)
    dt = time.perf_counter() - t0
    print(f"{algo} variant={variant} egress_wgs={wgs} {'u8' if u8 else 'f32'} pass {p}: {M / dt:8.1f} pairs/s ({dt*1e3:.1f} ms)", flush=True)
eng.close()
