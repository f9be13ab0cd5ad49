"""Mint tests/golden/prepare_golden.npz from the CPU oracle (oracle/prepare_oracle.c).

PARITY UNPINNED: cvtColor / cv::resize live in OpenCV, which is neither in /root/reference nor
installable here, so these are frozen outputs of the restatement (checked against the NumPy
restatement and known answers by tests/test_oracle_prepare.py).  Usage: python tests/golden/make_prepare_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [  # name, source w, h, channels, destination w, h
    ("down_340x256_to_224", 340, 256, 1, 224, 224),
    ("down_bgr_97x61_to_40x30", 97, 61, 3, 40, 30),
    ("half_128x96", 128, 96, 1, 64, 48),          # exact 2x: the INTER_AREA switch
    ("up_33x17_to_64x64", 33, 17, 1, 64, 64),
    ("gray_only_bgr_50x20", 50, 20, 3, 50, 20),    # same size: colour conversion only
]


def source(w, h, ch, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 90 * np.sin(xx / 5.3) * np.cos(yy / 4.1)
    planes = [np.clip(base + rng.normal(0, 25, (h, w)) + 20 * k, 0, 255) for k in range(ch)]
    img = np.stack(planes, -1).astype(np.uint8)
    return img[..., 0] if ch == 1 else img


def main():
    O.build()
    out = {}
    for i, (name, sw, sh, ch, dw, dh) in enumerate(CASES):
        src = source(sw, sh, ch, 40 + i)
        out[name + "_src"] = src
        out[name + "_dst"] = O.prepare_frame(src, dw, dh)
        print(name, out[name + "_dst"].shape, int(out[name + "_dst"].mean()))
    np.savez_compressed(os.path.join(HERE, "prepare_golden.npz"), **out)


if __name__ == "__main__":
    main()
