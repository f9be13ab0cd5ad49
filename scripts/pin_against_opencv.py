#!/usr/bin/env python3
"""Pin the oracle to real OpenCV output — to be run on ANY machine that has an OpenCV build with the CUDA modules
(`cv2.cuda`, opencv_contrib cudaoptflow; the reference pins 4.5.2, /root/reference/docker/Dockerfile:6).

This environment has no OpenCV, so the flow oracles are "parity unpinned".  This script is the kit that
closes the gap the moment such a machine is reachable: it evaluates the very calls the reference makes
(/root/reference/src/denseflow_gpu.cpp:299-303, :327-334) on the committed synthetic seeds and writes

    tests/golden/opencv_tvl1.npz   opencv_farn.npz   opencv_brox.npz   [opencv_cpu_tvl1.npz]

each holding, per case, the two uint8 frames and the CV_32FC2 flow OpenCV returned, plus cv2.getBuildInformation().
tests/test_opencv_pin.py picks the files up when present: the CPU oracle (-m "not gpu") and the HIP path (-m gpu)
must then agree with OpenCV within BASELINE.json's 1e-3 max-abs (TVL1, Farneback) — and the Brox oracle, which today
DEFINES its algorithm, gets its first external check.  Nothing here imports the product or the oracle.

    python scripts/pin_against_opencv.py [--out tests/golden] [--testdata $OPENCV_TEST_DATA_PATH]
With --testdata the upstream fixtures SURVEY.md §4 names are added as cases: cv/optflow/RubberWhale1.png / 2.png
(opencv_extra) and gpu/opticalflow/{frame0,frame1}.png with the stored Brox golden opticalflow/brox_optical_flow.bin."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from denseflow_amd.synth import SynthClip  # noqa: E402  (numpy only)

# (width, height, seed, t0, t1): the seeds of tests/golden/make_golden.py and SURVEY.md §8d
CASES = [(64, 48, 3, 0, 1), (224, 224, 1, 0, 1), (224, 224, 1, 3, 1), (320, 200, 6, 0, 2), (1920, 1080, 2, 0, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--testdata", default=os.environ.get("OPENCV_TEST_DATA_PATH"))
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        sys.exit("cv2 is not importable here: run this on a machine with OpenCV + opencv_contrib built WITH_CUDA")
    if not hasattr(cv2, "cuda") or cv2.cuda.getCudaEnabledDeviceCount() < 1:
        sys.exit("this OpenCV has no usable CUDA device: the reference's cv::cuda path cannot be evaluated")
    info = cv2.getBuildInformation()
    pairs = []
    for (w, h, seed, t0, t1) in CASES:
        clip = SynthClip(w, h, seed)
        pairs.append((f"synth_{w}x{h}_s{seed}_t{t0}_{t1}", clip.frame(t0), clip.frame(t1)))
    if args.testdata:
        for name, a, b in (("rubberwhale", "cv/optflow/RubberWhale1.png", "cv/optflow/RubberWhale2.png"),
                           ("gpu_frames", "gpu/opticalflow/frame0.png", "gpu/opticalflow/frame1.png")):
            fa, fb = (cv2.imread(os.path.join(args.testdata, p), cv2.IMREAD_GRAYSCALE) for p in (a, b))
            if fa is not None and fb is not None:
                pairs.append((name, fa, fb))

    def up(a):
        g = cv2.cuda_GpuMat()
        g.upload(a)
        return g

    algos = {
        "tvl1": lambda a, b: cv2.cuda_OpticalFlowDual_TVL1.create().calc(up(a), up(b), None).download(),
        "farn": lambda a, b: cv2.cuda_FarnebackOpticalFlow.create().calc(up(a), up(b), None).download(),
        # src/denseflow_gpu.cpp:303, :332-334: create(0.197f, 50.0f, 0.8f, 10, 77, 10) on frames scaled by 1/255
        "brox": lambda a, b: cv2.cuda_BroxOpticalFlow.create(0.197, 50.0, 0.8, 10, 77, 10).calc(
            up(a.astype(np.float32) * np.float32(1.0 / 255.0)), up(b.astype(np.float32) * np.float32(1.0 / 255.0)),
            None).download(),
    }
    os.makedirs(args.out, exist_ok=True)
    for algo, fn in algos.items():
        blob = {"build_information": np.array(info), "opencv_version": np.array(cv2.__version__)}
        for name, a, b in pairs:
            flow = fn(a, b)
            blob[name + "_f0"], blob[name + "_f1"], blob[name + "_flow"] = a, b, flow.astype(np.float32)
            print(algo, name, flow.shape, float(np.abs(flow).max()))
        np.savez_compressed(os.path.join(args.out, f"opencv_{algo}.npz"), **blob)
    if hasattr(cv2, "optflow") and hasattr(cv2.optflow, "DualTVL1OpticalFlow_create"):  # the CPU comparator
        blob = {"build_information": np.array(info), "opencv_version": np.array(cv2.__version__)}
        for name, a, b in pairs[:3]:
            blob[name + "_f0"], blob[name + "_f1"] = a, b
            blob[name + "_flow"] = cv2.optflow.DualTVL1OpticalFlow_create().calc(a, b, None).astype(np.float32)
        np.savez_compressed(os.path.join(args.out, "opencv_cpu_tvl1.npz"), **blob)
    if args.testdata:
        p = os.path.join(args.testdata, "gpu/opticalflow/brox_optical_flow.bin")
        if os.path.exists(p):
            print("upstream Brox golden present:", p, "(compare with opencv_brox.npz['gpu_frames_flow'])")


if __name__ == "__main__":
    main()
