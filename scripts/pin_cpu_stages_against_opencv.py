#!/usr/bin/env python3
"""Pin the stages the reference runs on the CPU — the save stage's encoders and the loader's gray conversion + resize — to
OpenCV's own output.  Runs on ANY machine with a plain `cv2` (no CUDA needed; e.g. `pip install opencv-python`).  The JPEG / PNG writers of this repository are already held to libjpeg-turbo's and libpng's
files (tests/test_jpeg_libjpeg_pin.py, tests/test_png_libpng_pin.py), on the understanding that cv::imencode drives those
libraries with their defaults (JPEG) and with SUB filter + Z_BEST_SPEED + Z_RLE (PNG): this script replaces that
understanding with cv2's output.  It encodes the planes / images of tests/golden/jpeg_golden.npz and png_golden.npz with

    cv2.imencode(".jpg", plane)      (the reference's call, /root/reference/src/common.cpp:56-57: default quality 95)
    cv2.imencode(".png", bgr)        (/root/reference/src/common.cpp:70)

and writes tests/golden/opencv_imencode.npz (files + cv2.getBuildInformation()); tests/test_opencv_pin.py::
test_encoders_reproduce_cv2_imencode then holds the host encoders (and, on the GPU box, the device JPEG encoder) to those
bytes.

Frame preparation ("parity unpinned" today): the reference's loader calls, on the CPU,

    cvtColor(frame, gray, COLOR_BGR2GRAY)     (/root/reference/src/denseflow_gpu.cpp:163)
    cv::resize(gray, resized, size)            (:169, default INTER_LINEAR)

This script applies exactly those to the sources of tests/golden/prepare_golden.npz and writes
tests/golden/opencv_prepare.npz; test_frame_preparation_reproduces_cv2 then holds the oracle (CPU) and the device kernel
(GPU) to cv2's pixels, bit for bit.

The CPU comparator of bench.py (`cpu_baseline`: cv::optflow::DualTVL1OpticalFlow restated in oracle/cpu_tvl1_baseline.c): when
the installed cv2 has the contrib modules (`pip install opencv-contrib-python`) the same three synthetic pairs as
scripts/pin_against_opencv.py uses are run through cv2.optflow.DualTVL1OpticalFlow_create().calc and written to
tests/golden/opencv_cpu_tvl1.npz (tests/test_opencv_pin.py::test_cpu_baseline_port_reproduces_opencv_cpu_dualtvl1).
Nothing here imports the product.

    python scripts/pin_cpu_stages_against_opencv.py [--out tests/golden]"""
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

    prep = {"version": np.frombuffer(cv2.__version__.encode(), np.uint8)}
    g = np.load(os.path.join(ROOT, "tests", "golden", "prepare_golden.npz"))
    for k in g.files:
        if not k.endswith("_src"):
            continue
        src, want = np.ascontiguousarray(g[k]), g[k[:-4] + "_dst"]
        gray = cv2.cvtColor(src, cv2.COLOR_BGR2GRAY) if src.ndim == 3 else src
        dh, dw = want.shape
        out = cv2.resize(gray, (dw, dh)) if gray.shape != (dh, dw) else gray
        prep[k[:-4]] = np.ascontiguousarray(out)
        print("prepare", k[:-4], src.shape, "->", out.shape, "differs from the committed oracle golden in",
              int(np.count_nonzero(out != want)), "pixels")
    np.savez_compressed(os.path.join(args.out, "opencv_prepare.npz"), **prep)
    print("wrote", os.path.join(args.out, "opencv_prepare.npz"))

    sys.path.insert(0, ROOT)
    from denseflow_amd.synth import SynthClip  # numpy only

    # CROSS-CHECK, not a pin: CPU cv2.calcOpticalFlowFarneback with the parameters cv::cuda::FarnebackOpticalFlow::create()
    # defaults to (5 levels, 0.5, window 13, 10 iterations, polyN 5, sigma 1.1, flags 0).  The CUDA class is a port of this
    # function with its own border and blur details, so the flows agree closely, not to 1e-3; a large gap to the oracle
    # would still expose a structural misreading of Appendix B.
    fb = {"opencv_version": np.array(cv2.__version__)}
    for (w, h, seed, t0, t1) in [(64, 48, 3, 0, 1), (224, 224, 1, 0, 1), (320, 200, 6, 0, 2)]:
        clip = SynthClip(w, h, seed)
        name = f"synth_{w}x{h}_s{seed}_t{t0}_{t1}"
        a, b = clip.frame(t0), clip.frame(t1)
        fb[name + "_f0"], fb[name + "_f1"] = a, b
        fb[name + "_flow"] = cv2.calcOpticalFlowFarneback(a, b, None, 0.5, 5, 13, 10, 5, 1.1, 0).astype(np.float32)
        print("cpu Farneback", name, float(np.abs(fb[name + "_flow"]).max()))
    np.savez_compressed(os.path.join(args.out, "opencv_cpu_farneback.npz"), **fb)
    print("wrote", os.path.join(args.out, "opencv_cpu_farneback.npz"))

    if hasattr(cv2, "optflow") and hasattr(cv2.optflow, "DualTVL1OpticalFlow_create"):

        tv = {"opencv_version": np.array(cv2.__version__)}
        for (w, h, seed, t0, t1) in [(64, 48, 3, 0, 1), (224, 224, 1, 0, 1), (224, 224, 1, 3, 1)]:  # pin_against_opencv.py's
            clip = SynthClip(w, h, seed)
            name = f"synth_{w}x{h}_s{seed}_t{t0}_{t1}"
            a, b = clip.frame(t0), clip.frame(t1)
            tv[name + "_f0"], tv[name + "_f1"] = a, b
            tv[name + "_flow"] = cv2.optflow.DualTVL1OpticalFlow_create().calc(a, b, None).astype(np.float32)
            print("cpu DualTVL1", name, float(np.abs(tv[name + "_flow"]).max()))
        np.savez_compressed(os.path.join(args.out, "opencv_cpu_tvl1.npz"), **tv)
        print("wrote", os.path.join(args.out, "opencv_cpu_tvl1.npz"))
    else:
        print("cv2.optflow absent (plain opencv-python): the CPU DualTVL1 comparator stays unpinned; "
              "opencv-contrib-python has it")


if __name__ == "__main__":
    main()
