#!/usr/bin/env python3
"""Mint tests/golden/png_golden.npz: images and the PNG files the system's **libpng** (1.6.37 + zlib 1.2.11 here) writes
for them under the calls cv::imencode(".png") makes (tests/libpng_ref.py: SUB filter, Z_BEST_SPEED, Z_RLE, png_set_bgr)
— the last step of the reference's encodeFlowMapPng, /root/reference/src/common.cpp:70.  Outputs of the real library,
not of this repository's writer; tests/test_png_libpng_pin.py holds the shell's imencodePng to these bytes where no
libpng can be loaded, and to the live library where it can.
    python tests/golden/make_png_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = {"gray_1x1": (1, 1, 1), "gray_1x9": (1, 9, 1), "bgr_9x1": (9, 1, 3), "bgr_45x27": (45, 27, 3), "gray_96x64": (96, 64, 1),
         "bgr_74x74": (74, 74, 3), "bgr_noise_64x48": (64, 48, 3), "bgr_flowlike_320x180": (320, 180, 3)}


def images():
    rng = np.random.default_rng(77)
    out = {}
    for name, (w, h, ch) in CASES.items():
        yy, xx = np.mgrid[0:h, 0:w]
        if "noise" in name:
            img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        elif "flowlike" in name:  # what convertFlowToPngImage produces: two smooth planes + a constant-per-row third
            a = 128 + 50 * np.exp(-((xx - 100) ** 2 + (yy - 60) ** 2) / 900.0) + rng.normal(0, 0.5, (h, w))
            b = 128 - 40 * np.exp(-((xx - 220) ** 2 + (yy - 120) ** 2) / 1600.0)
            c = np.repeat((np.arange(h) * 7 % 256)[:, None], w, 1)
            img = np.clip(np.rint(np.stack([a, b, c], -1)), 0, 255).astype(np.uint8)
        else:
            base = 128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
            img = np.clip(np.rint(np.stack([base, 255 - base, base / 2], -1)[..., :ch]), 0, 255).astype(np.uint8)
        out[name] = img[..., 0] if ch == 1 else img
    return out


def main():
    from tests.libpng_ref import imencode_png, load

    L = load()
    blob = {"libpng": np.frombuffer(b"libpng " + L.png_get_libpng_ver(None), np.uint8)}
    for name, img in images().items():
        blob[name + "_image"] = img
        blob[name + "_file"] = np.frombuffer(imencode_png(img), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "png_golden.npz"), **blob)
    print({k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
