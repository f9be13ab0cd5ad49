"""Mint tests/golden/png_planes_golden.npz from THE REFERENCE'S OWN CODE.

convertFlowToPngImage (/root/reference/src/common.cpp:18-46, the -st=png scheme) is compiled as it stands by
`make -C oracle ref` into oracle/_ref/libref_png.so (oracle/ref_png_shim.h stands in for cv::Mat / minMaxLoc /
convertTo / rectangle / mixChannels) and run here on flows built to sit on the branches of the bound rule — maxima that
land bound on a multiple of 8 (the `+= 4` step) and next to it, maxima beyond the frame size (the min(w, .)), beyond
1020 (the min(255*4, .)), all-zero flows — and on the rounding ties of the two convertTo planes.  Needs /root/reference;
the resulting file travels with the repository.  Usage:  python tests/golden/make_png_planes_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def flow_with_extrema(w, h, max_u, max_v, seed, neg_u=False):
    """(h, w, 2) float32 whose largest |u| is exactly max_u (negative if neg_u) and largest |v| exactly max_v; the rest
    noise, ties of the convertTo rounding (k + 0.5 after scaling) and their float neighbours."""
    rng = np.random.default_rng(seed)
    flow = (rng.uniform(-1, 1, (h, w, 2)) * np.array([max_u, max_v]) * 0.999).astype(np.float32)
    for c, m in ((0, max_u), (1, max_v)):
        if m > 0:
            extent = w if c == 0 else h
            bound = min(1020.0, np.ceil((min(extent, m) * 128.0 / 127.0) / 4) * 4)
            if int(bound) % 8 == 0:
                bound += 4
            k = np.arange(-120, 121, 7, dtype=np.float64) + 0.5  # (v / (bound/128)) + 128 = tie
            ties = (k * bound / 128.0).astype(np.float32)
            ties = ties[np.abs(ties) < m * 0.999]
            pool = np.concatenate([ties, np.nextafter(ties, np.float32(np.inf)), np.nextafter(ties, np.float32(-np.inf))])
            idx = rng.choice(h * w, size=min(pool.size, h * w // 2), replace=False)
            flow.reshape(-1, 2)[idx, c] = pool[: idx.size]
    flow[h // 3, w // 3, 0] = -max_u if neg_u else max_u
    flow[h // 2, w // 2, 1] = max_v
    return flow


CASES = [  # name, w, h, max|u|, max|v|, seed, max u negative
    ("small", 96, 64, 3.2, 1.1, 1, False),
    ("mult8", 64, 48, 15.5, 7.9, 2, True),       # ceil(15.62/4)*4 = 16 -> % 8 == 0 -> 20; 7.96 -> 8 -> 12
    ("next_to_mult8", 61, 37, 11.9, 19.8, 3, False),  # 12, 20: no += 4
    ("beyond_frame", 40, 30, 500.0, 77.0, 4, False),  # min(w, .) = 40 -> 44; min(h, .) = 30 -> 32 -> 36
    ("beyond_1020", 2000, 8, 1500.0, 2.0, 5, True),   # 1511.8 -> min(1020, 1512) = 1020 (1020 % 8 = 4)
    ("zero", 33, 31, 0.0, 0.0, 6, False),             # bound 0 -> % 8 == 0 -> 4
    ("odd_rows", 50, 41, 6.0, 2.5, 7, False),         # int(h / 2) = 20: rows 0..20 carry x's bound
]


def main():
    O.build()
    assert O.ref_png_available(), "oracle/_ref/libref_png.so missing (needs /root/reference)"
    out = {}
    for name, w, h, mu, mv, seed, neg in CASES:
        flow = flow_with_extrema(w, h, mu, mv, seed, neg)
        bgr = O.ref_flow_to_png_image(flow)
        out[name + "_flow"] = flow
        out[name + "_bgr"] = bgr
        print(name, "channel 2 values", np.unique(bgr[..., 2]).tolist())
    np.savez_compressed(os.path.join(HERE, "png_planes_golden.npz"), **out)


if __name__ == "__main__":
    main()
