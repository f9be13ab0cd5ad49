"""Mint tests/golden/quant_golden.npz from THE REFERENCE'S OWN CODE.

Unlike the optical-flow goldens, these vectors are pinned: convertFlowToImage
(/root/reference/src/common.cpp:4-16) is compiled as it stands by `make -C oracle ref` into
oracle/_ref/libref_quant.so (a cv::Mat / cvRound stand-in is the only thing added) and run here on
inputs built to sit on every branch and on the rounding ties.  Needs /root/reference; the resulting
file travels with the repository.  Usage:  python tests/golden/make_quant_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def adversarial_flow(bound: float, w: int, h: int, seed: int) -> np.ndarray:
    """(h, w, 2) float32: rounding ties k + 0.5, their float neighbours, the bounds, specials, noise."""
    rng = np.random.default_rng(seed)
    k = np.arange(0, 255, dtype=np.float64) + 0.5
    ties = (k * (2 * bound) / 255 - bound).astype(np.float32)  # 255*(v+b)/(2b) ~ k + 0.5
    near = np.concatenate([ties, np.nextafter(ties, np.float32(np.inf)), np.nextafter(ties, np.float32(-np.inf))])
    b32 = np.float32(bound)
    special = np.array([0.0, -0.0, b32, -b32, np.nextafter(b32, np.float32(np.inf)), np.nextafter(-b32, np.float32(-np.inf)),
                        np.nextafter(b32, np.float32(0)), np.nextafter(-b32, np.float32(0)), np.inf, -np.inf, np.nan, 1e30,
                        -1e30, 1e-40, -1e-40, 3.4e38, -3.4e38], np.float32)
    pool = np.concatenate([near, special, (rng.standard_normal(1024) * bound / 2).astype(np.float32)])
    flow = rng.choice(pool, size=(h, w, 2)).astype(np.float32)
    assert pool.size <= flow.size
    flow.reshape(-1)[: pool.size] = pool  # every pool value appears at least once
    return flow


CASES = [("b20", 20.0, 96, 64, 1), ("b32", 32.0, 61, 37, 2), ("b15", 15.0, 130, 33, 3), ("b1", 1.0, 64, 64, 4)]


def main():
    O.build()
    assert O.ref_quant_available(), "oracle/_ref/libref_quant.so missing (needs /root/reference)"
    out = {}
    for name, bound, w, h, seed in CASES:
        flow = adversarial_flow(bound, w, h, seed)
        ix, iy = O.ref_flow_to_u8(flow, -bound, bound)
        out[name + "_bound"] = np.array([bound])
        out[name + "_flow"] = flow
        out[name + "_x"] = ix
        out[name + "_y"] = iy
        print(name, "histogram ends", int((ix == 0).sum()), int((ix == 255).sum()))
    np.savez_compressed(os.path.join(HERE, "quant_golden.npz"), **out)


if __name__ == "__main__":
    main()
