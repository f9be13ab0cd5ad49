#!/usr/bin/env python3
"""PCIe-inclusive rate: host (page-locked) frames in, host flows out through dfx_calc_batch — the path the
host shell uses — next to the HBM-resident rate bench.py reports.  Usage: python scripts/pcie_rate.py [algo] [W H NF]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import denseflow_amd  # noqa: E402
from denseflow_amd.synth import SynthClip  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else "tvl1"
W, H, NF = (int(v) for v in (sys.argv[2:5] if len(sys.argv) >= 5 else (1920, 1080, 97)))
dev = torch.device("cuda", 0)
d_frames = SynthClip(W, H, 2).frames_torch(NF, dev)
h_frames = torch.empty((NF, H, W), dtype=torch.uint8, pin_memory=True)
h_frames.copy_(d_frames)
h_flows = torch.empty((NF - 1, H, W, 2), dtype=torch.float32, pin_memory=True)
d_flows = torch.empty((NF - 1, H, W, 2), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
eng = denseflow_amd.FlowEngine(W, H, algo)
L = denseflow_amd.load_library()
fp = (C.c_void_p * NF)(*[h_frames[i].data_ptr() for i in range(NF)])
op = (C.c_void_p * (NF - 1))(*[h_flows[i].data_ptr() for i in range(NF - 1)])


def host_path():
    rc = L.dfx_calc_batch(eng._h, fp, W, NF, 1, op, W * 8)
    assert rc == 0, L.dfx_last_error(eng._h)


h_x = torch.empty((NF - 1, H, W), dtype=torch.uint8, pin_memory=True)
h_y = torch.empty((NF - 1, H, W), dtype=torch.uint8, pin_memory=True)
xp = (C.c_void_p * (NF - 1))(*[h_x[i].data_ptr() for i in range(NF - 1)])
yp = (C.c_void_p * (NF - 1))(*[h_y[i].data_ptr() for i in range(NF - 1)])


def host_path_u8():  # flows bounded to [-20, 20] on the device: 2 B/px come down instead of 8
    rc = L.dfx_calc_batch_u8(eng._h, fp, W, NF, 1, -20.0, 20.0, xp, yp, W)
    assert rc == 0, L.dfx_last_error(eng._h)


def device_path():
    eng.calc_optflows_device(d_frames.data_ptr(), W, W * H, NF, 1, d_flows.data_ptr(), W * H * 2)


for name, fn in (("HBM-resident", device_path), ("PCIe-inclusive (pinned host in/out)", host_path),
                 ("PCIe-inclusive, bounded 8-bit planes out", host_path_u8)):
    fn()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print(f"{algo} {W}x{H}: {name}: {(NF - 1) / dt:.1f} pairs/s", flush=True)
torch.cuda.synchronize()
print("host == device results:", bool(torch.equal(h_flows, d_flows.cpu())))
