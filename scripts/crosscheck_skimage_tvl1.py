#!/usr/bin/env python3
"""Sanity cross-check of the TVL1 oracle against an INDEPENDENT third-party TV-L1 solver that happens to be on this
machine: scikit-image 0.18 (`skimage.registration.optical_flow_tvl1`, the IPOL / Wedel et al. formulation) under
/opt/conda/bin/python3.9.  It is NOT OpenCV's algorithm (other pyramid, other parameterisation, no dual variable p), so
this pins nothing: it only shows that the oracle computes a TV-L1 flow of the same quality on the committed seeds — both
are compared with the synthetic clip's true flow and with each other.
Usage: /opt/conda/bin/python3.9 scripts/crosscheck_skimage_tvl1.py  ->  markdown on stdout."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from skimage.registration import optical_flow_tvl1  # noqa: E402

from denseflow_amd.synth import SynthClip  # noqa: E402
from oracle import oracle_py  # noqa: E402

print("| frame size, seed, pair | oracle: mean / max end-point error vs truth (interior) | scikit-image TV-L1: same | oracle vs scikit-image: mean abs difference |")
print("|---|---|---|---|")
for (w, h, seed, t0, t1) in [(224, 224, 1, 0, 1), (224, 224, 2, 0, 1), (224, 224, 3, 0, 2), (320, 240, 5, 0, 1), (640, 360, 4, 0, 1)]:
    clip = SynthClip(w, h, seed)
    f0, f1 = clip.frame(t0), clip.frame(t1)
    gt = clip.true_flow(t0, t1)
    ours = oracle_py.tvl1_calc(f0, f1)
    v, u = optical_flow_tvl1(f0.astype(np.float32) / 255.0, f1.astype(np.float32) / 255.0)
    sk = np.stack([u, v], axis=-1).astype(np.float32)
    m = 16
    def epe(a):
        d = np.sqrt(((a - gt) ** 2).sum(-1))[m:-m, m:-m]
        return d.mean(), d.max()
    eo, es = epe(ours), epe(sk)
    dd = np.abs(ours - sk)[m:-m, m:-m].mean()
    print(f"| {w}x{h}, seed {seed}, {t0}->{t1} | {eo[0]:.4f} / {eo[1]:.3f} px | {es[0]:.4f} / {es[1]:.3f} px | {dd:.4f} px |")
