#!/usr/bin/env python3
"""Pin the save stage's encoders to OpenCV's own bytes — runs on ANY machine with a plain `cv2` (no CUDA needed; e.g.
`pip install opencv-python`).  The JPEG / PNG writers of this repository are already held to libjpeg-turbo's and libpng's
files (tests/test_jpeg_libjpeg_pin.py, tests/test_png_libpng_pin.py), on the understanding that cv::imencode drives those
libraries with their defaults (JPEG) and with SUB filter + Z_BEST_SPEED + Z_RLE (PNG): this script replaces that
understanding with cv2's output.  It encodes the planes / images of tests/golden/jpeg_golden.npz and png_golden.npz with

    cv2.imencode(".jpg", plane)      (the reference's call, /root/reference/src/common.cpp:56-57: default quality 95)
    cv2.imencode(".png", bgr)        (/root/reference/src/common.cpp:70)

and writes tests/golden/opencv_imencode.npz (files + cv2.getBuildInformation()); tests/test_opencv_pin.py::
test_encoders_reproduce_cv2_imencode then holds the host encoders (and, on the GPU box, the device JPEG encoder) to those
bytes.  Nothing here imports the product.

    python scripts/pin_imencode_against_opencv.py [--out tests/golden]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        sys.exit("cv2 is not importable here: any OpenCV Python package will do (no CUDA needed)")
    blob = {"build_information": np.frombuffer(cv2.getBuildInformation().encode(), np.uint8),
            "version": np.frombuffer(cv2.__version__.encode(), np.uint8)}
    jg = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_golden.npz"))
    for k in jg.files:
        if k.endswith("_plane"):
            ok, buf = cv2.imencode(".jpg", np.ascontiguousarray(jg[k]))
            assert ok
            blob["jpg_" + k[:-6]] = np.asarray(buf, np.uint8).ravel()
            print("jpg", k[:-6], jg[k].shape, blob["jpg_" + k[:-6]].size, "bytes")
    pg = np.load(os.path.join(ROOT, "tests", "golden", "png_golden.npz"))
    for k in pg.files:
        if k.endswith("_image"):
            ok, buf = cv2.imencode(".png", np.ascontiguousarray(pg[k]))
            assert ok
            blob["png_" + k[:-6]] = np.asarray(buf, np.uint8).ravel()
            print("png", k[:-6], pg[k].shape, blob["png_" + k[:-6]].size, "bytes")
    np.savez_compressed(os.path.join(args.out, "opencv_imencode.npz"), **blob)
    print("wrote", os.path.join(args.out, "opencv_imencode.npz"), "with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()
