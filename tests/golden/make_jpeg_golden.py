#!/usr/bin/env python3
"""Mint tests/golden/jpeg_golden.npz: planes and the JPEG files the shell's host encoder (src/image_io.cpp,
imencodeJpeg; shared tables include/dfx_jpeg_tables.h) writes for them.  The host encoder (CPU suite) and the device
encoder (GPU suite, dfx_calc_batch_jpeg on flows whose bounded planes are these) are both held to these bytes, so an
accidental change of either encoder's arithmetic or tables shows up even where the two would still agree with each other.
    python tests/golden/make_jpeg_golden.py        (needs `make host`; CPU only)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def harness():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhost_harness.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", so,
           os.path.join(ROOT, "tests", "host_harness.cpp"), os.path.join(ROOT, "build", "libzzdenseflow.a"),
           "-L" + os.path.join(ROOT, "denseflow_amd", "lib"), "-ldfx", "-lpthread", "-lz",
           "-Wl,-rpath," + os.path.join(ROOT, "denseflow_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return C.CDLL(so)


def planes():
    rng = np.random.default_rng(2026)
    out = {}
    for name, (w, h) in {"smooth_96x64": (96, 64), "ragged_45x27": (45, 27), "busy_64x64": (64, 64)}.items():
        yy, xx = np.mgrid[0:h, 0:w]
        p = 128 + 70 * np.sin(xx / 13.0 + 0.3) * np.cos(yy / 9.0) + (rng.normal(0, 30, (h, w)) if "busy" in name else 0)
        p = np.clip(p, 0, 255).astype(np.uint8)
        if "ragged" in name:
            p[:9, :11] = 255  # a saturated corner: 0xFF bytes in the segment (byte stuffing)
        out[name] = p
    return out


def main():
    H = harness()
    H.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    blob = {}
    for name, p in planes().items():
        for q in (95, 50):
            buf = np.zeros(1 << 20, np.uint8)
            n = H.hh_encode_jpeg(np.ascontiguousarray(p).ctypes.data, p.shape[1], p.shape[0], q, buf.ctypes.data, buf.size)
            assert n > 0
            blob[f"{name}_q{q}_file"] = buf[:n].copy()
        blob[name + "_plane"] = p
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_golden.npz"), **blob)
    print({k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
