#!/usr/bin/env python3
"""Mint tests/golden/jpeg_golden.npz: gray planes and the JPEG files **libjpeg-turbo** writes for them (through Pillow,
`Image.save(..., "JPEG", quality=q)`: libjpeg's defaults — baseline, Annex K tables, JDCT_ISLOW — which is how
cv::imencode(".jpg") of the reference's encodeFlowMap, /root/reference/src/common.cpp:56-57, drives the same library).
libjpeg is a third-party dependency of the reference that is not in /root/reference; Pillow's copy is importable in this
container, so these are outputs of the real thing, not of this repository's encoders.  The host encoder (CPU suite,
tests/test_jpeg_host.py, tests/test_jpeg_libjpeg_pin.py) and the device encoder (GPU suite, tests/test_jpeg_gpu.py) are
both held to these bytes.
    python tests/golden/make_jpeg_golden.py        (needs Pillow; CPU only)"""
import io
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
QUALITIES = (95, 50, 100, 10)
CASES = {"smooth_96x64": (96, 64), "ragged_45x27": (45, 27), "busy_64x64": (64, 64), "noise_40x24": (40, 24),
         "tiny_5x3": (5, 3), "one_1x1": (1, 1), "flowlike_257x131": (257, 131)}


def planes():
    rng = np.random.default_rng(2026)
    out = {}
    for name, (w, h) in CASES.items():
        yy, xx = np.mgrid[0:h, 0:w]
        if "noise" in name:
            p = rng.integers(0, 256, (h, w)).astype(np.float64)
        elif "flowlike" in name:  # what a bounded flow plane looks like: smooth, near 128, a few moving blobs
            p = 128 + 40 * np.exp(-((xx - 90) ** 2 + (yy - 50) ** 2) / 900.0) - 55 * np.exp(-((xx - 190) ** 2 + (yy - 90) ** 2) / 400.0)
            p = p + rng.normal(0, 0.6, (h, w))
        else:
            p = 128 + 70 * np.sin(xx / 13.0 + 0.3) * np.cos(yy / 9.0) + (rng.normal(0, 30, (h, w)) if "busy" in name else 0)
        p = np.clip(np.rint(p), 0, 255).astype(np.uint8)
        if "ragged" in name:
            p[:9, :11] = 255  # a saturated corner: 0xFF bytes in the segment (byte stuffing)
        out[name] = p
    return out


def libjpeg_file(plane, quality):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(plane), "L").save(b, "JPEG", quality=quality)
    return b.getvalue()


def main():
    from PIL import features

    blob = {"libjpeg": np.frombuffer(f"libjpeg-turbo {features.version('jpg')} (Pillow)".encode(), np.uint8)}
    for name, p in planes().items():
        for q in QUALITIES:
            blob[f"{name}_q{q}_file"] = np.frombuffer(libjpeg_file(p, q), np.uint8)
        blob[name + "_plane"] = p
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_golden.npz"), **blob)
    print({k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
