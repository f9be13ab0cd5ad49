"""What buffer addressing does on this device (dfxi_probe_buffer, selftest.hip): run on the GPU box."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import denseflow_amd as dfx

lib = dfx.load_library()
fn = lib.dfxi_probe_buffer
fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
fn.restype = C.c_int
n = 1000
x = np.arange(1, n + 1, dtype=np.float32)
for mode in range(4):
    y = np.full(n, mode, np.float32)
    out = np.empty_like(x)
    assert fn(0, x.ctypes.data, y.ctypes.data, out.ctypes.data, n) == 0
    want = x + 1 if mode == 3 else x
    bad = np.flatnonzero(out != want)
    print("mode", mode, "mismatches", bad.size, "first", bad[:6].tolist(), "values", out[bad[:6]].tolist(), "last 6 out", out[-6:].tolist())
