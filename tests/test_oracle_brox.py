"""CPU tests of the Brox oracle (oracle/brox_oracle.c).  PARITY UNPINNED and spec confidence LOW
(SURVEY.md Appendix C): the algorithm is DEFINED in oracle/brox_oracle.h; these tests pin the C
implementation of that definition against an independent NumPy implementation, analytic answers and
frozen goldens."""
import os

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip
from tests import numpy_restatement as NR

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_pyramid_sizes(oracle):
    assert oracle.brox_pyramid_sizes(224, 224)[:4] == [(224, 224), (180, 180), (144, 144), (115, 115)]
    assert oracle.brox_pyramid_sizes(224, 224) == NR.brox_pyramid_sizes(224, 224)
    s4k = oracle.brox_pyramid_sizes(3840, 2160)
    assert len(s4k) == 24 and min(s4k[-1]) <= 15 and min(s4k[-2]) > 15  # ~23 levels below full res
    p = oracle.brox_default_params()
    p.outer_iterations = 3
    assert len(oracle.brox_pyramid_sizes(640, 480, p)) == 3


def test_zero_motion_is_exactly_zero(oracle):
    f = SynthClip(120, 90, 3).frame(0)
    assert np.all(oracle.brox_calc(f, f) == 0.0)  # Iz = Ixz = Iyz = 0 -> num = 0 -> du = dv = 0 throughout


def test_translation_recovered(oracle):
    clip = SynthClip(200, 150, 4)
    flow = oracle.brox_calc(clip.frame(0), clip.frame(1))
    gt = clip.true_flow(0, 1)
    err = np.abs(flow - gt)[16:-16, 16:-16]
    assert np.isfinite(flow).all() and err.mean() < 0.03 and err.max() < 0.3


def test_oracle_matches_numpy_restatement(oracle):
    w, h = 40, 32
    clip = SynthClip(w, h, 6)
    f0, f1 = clip.frame(0), clip.frame(1)
    a = oracle.brox_calc(f0, f1)
    b = NR.brox_calc(f0, f1)
    assert np.max(np.abs(a - b)) <= 1e-5


def test_golden_vectors(oracle):
    g = np.load(os.path.join(GOLDEN, "brox_golden.npz"))
    for key in [k[:-5] for k in g.files if k.endswith("_flow")]:
        w, h, seed, t0, t1 = [int(v) for v in g[key + "_meta"]]
        clip = SynthClip(w, h, seed)
        assert np.array_equal(oracle.brox_calc(clip.frame(t0), clip.frame(t1)), g[key + "_flow"]), key
