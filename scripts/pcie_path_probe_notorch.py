#!/usr/bin/env python3
"""The PCIe-inclusive path in a process WITHOUT torch: libdfx.so then runs on the system HIP runtime it was linked against
(/opt/rocm, 7.2) — what a C++ caller such as build/denseflow gets — instead of the HIP 7.0 runtime torch bundles.  The two
execute device-to-host copies differently (7.0: shader blit `__amd_rocclr_copyBuffer`; 7.2: SDMA), which is what
decides the float-output rate of an HBM-bound algorithm.
    step 1 (needs torch, any process):  python scripts/pcie_path_probe_notorch.py make W H NF /tmp/clip.npy
    step 2:  DFX_NO_TORCH=1 python scripts/pcie_path_probe_notorch.py run ALGO /tmp/clip.npy [PASSES]
Prints pairs/s for float flows, bounded planes and JPEG files out (synchronous calls), and for two FlowBuffers in flight."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

if sys.argv[1] == "make":
    import torch

    from denseflow_amd.synth import SynthClip

    W, H, NF = (int(v) for v in sys.argv[2:5])
    np.save(sys.argv[5], SynthClip(W, H, 2).frames_torch(NF, torch.device("cuda", 0)).cpu().numpy())
    sys.exit(0)

assert os.environ.get("DFX_NO_TORCH") == "1" and "torch" not in sys.modules
import denseflow_amd  # noqa: E402

algo, clip = sys.argv[2], np.load(sys.argv[3])
PASSES = int(sys.argv[4]) if len(sys.argv) > 4 else 3
assert "torch" not in sys.modules
NF, H, W = clip.shape
M = NF - 1
L = denseflow_amd.load_library()
with open("/proc/self/maps") as f:
    print("HIP runtime:", sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln}))


def pinned(nbytes):
    p = C.c_void_p()
    assert L.dfx_host_alloc(C.byref(p), nbytes) == 0
    return p.value


fb = pinned(NF * H * W)
C.memmove(fb, clip.ctypes.data, NF * H * W)
fp = (C.c_void_p * NF)(*[fb + i * H * W for i in range(NF)])
eng = denseflow_amd.FlowEngine(W, H, algo)
cap = int(L.dfx_jpeg_capacity(eng._h))
flows = pinned(M * H * W * 8)
op = (C.c_void_p * M)(*[flows + i * H * W * 8 for i in range(M)])
planes = [pinned(M * H * W) for _ in range(4)]
pp = [(C.c_void_p * M)(*[b + i * H * W for i in range(M)]) for b in planes]
jbuf = [pinned(M * cap) for _ in range(4)]
jp = [(C.c_void_p * M)(*[b + i * cap for i in range(M)]) for b in jbuf]
js = [(C.c_uint32 * M)() for _ in range(4)]


def check(rc):
    assert rc == 0, L.dfx_last_error(eng._h)


legs = {
    "float flows out": lambda: check(L.dfx_calc_batch(eng._h, fp, W, NF, 1, op, W * 8)),
    "bounded planes out": lambda: check(L.dfx_calc_batch_u8(eng._h, fp, W, NF, 1, -20.0, 20.0, pp[0], pp[1], W)),
    "JPEG files out": lambda: check(L.dfx_calc_batch_jpeg(eng._h, fp, W, NF, 1, -20.0, 20.0, 95, jp[0], jp[1], cap, js[0], js[1])),
}
for name, fn in legs.items():
    fn()
    best = 0.0
    for _ in range(PASSES):
        t0 = time.perf_counter()
        fn()
        best = max(best, M / (time.perf_counter() - t0))
    print(f"{algo} {W}x{H} {name}: {best:8.1f} pairs/s", flush=True)


def in_flight(kind, n_fb=4):
    tickets = []
    for k in range(n_fb):
        if k >= 2:
            check(L.dfx_wait(eng._h, tickets[k - 2]))
        t = C.c_uint64(0)
        s = k & 1
        if kind == "jpeg":
            check(L.dfx_submit_batch_jpeg(eng._h, fp, W, NF, 1, -20.0, 20.0, 95, jp[2 * s], jp[2 * s + 1], cap, js[2 * s], js[2 * s + 1], C.byref(t)))
        elif kind == "u8":
            check(L.dfx_submit_batch_u8(eng._h, fp, W, NF, 1, -20.0, 20.0, pp[2 * s], pp[2 * s + 1], W, C.byref(t)))
        else:
            check(L.dfx_submit_batch(eng._h, fp, W, NF, 1, op, W * 8, C.byref(t)))  # one output set: results are overwritten
        tickets.append(t.value)
    check(L.dfx_wait(eng._h, 0))


for kind in ("f32", "u8", "jpeg"):
    in_flight(kind, 2)
    t0 = time.perf_counter()
    in_flight(kind, 4)
    print(f"{algo} {W}x{H} {kind} with FlowBuffers in flight: {4 * M / (time.perf_counter() - t0):8.1f} pairs/s", flush=True)
eng.close()
