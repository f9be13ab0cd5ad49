"""CPU tests of the Farneback oracle (oracle/farneback_oracle.c).  PARITY UNPINNED (no reference
vectors exist); pinned against the independent NumPy restatement, sanity answers and frozen goldens."""
import os

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip
from tests import numpy_restatement as NR

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_translation_recovered_in_interior(oracle):
    clip = SynthClip(256, 192, 12)
    flow = oracle.farneback_calc(clip.frame(0), clip.frame(1))
    gt = clip.true_flow(0, 1)
    b = 24
    err = np.abs(flow - gt)[b:-b, b:-b]
    assert err.mean() < 0.05 and err.max() < 0.5  # sanity, not parity


def test_identical_frames_give_small_border_driven_flow(oracle):
    """Upstream quirk (B.7 else-branch): at the last row/column R1 is not sampled, so h != 0 there even
    for identical frames; the flow is NOT exactly zero (SURVEY.md §8c KAT 1 does not hold for Farneback)."""
    f = SynthClip(256, 192, 3).frame(0)
    z = oracle.farneback_calc(f, f)
    assert 0 < np.abs(z).max() < 1.0
    assert np.abs(z)[40:-60, 40:-60].mean() < 0.05


@pytest.mark.parametrize("w,h,seed", [(64, 48, 3), (130, 97, 5), (224, 224, 1)])
def test_oracle_matches_numpy_restatement(oracle, w, h, seed):
    clip = SynthClip(w, h, seed)
    f0, f1 = clip.frame(0), clip.frame(1)
    a = oracle.farneback_calc(f0, f1)
    b = NR.farneback_calc(f0, f1)
    assert np.max(np.abs(a - b)) <= 2e-4  # same op order; only the Gaussian taps / matrix inverse differ in the last ulp


def test_constants(oracle):
    import ctypes as C

    class PC(C.Structure):
        _fields_ = [("g", C.c_float * 8), ("xg", C.c_float * 8), ("xxg", C.c_float * 8), ("ig11", C.c_float),
                    ("ig03", C.c_float), ("ig33", C.c_float), ("ig55", C.c_float)]

    pc = PC()
    L = oracle.lib()
    L.orc_farneback_prepare_poly.argtypes = [C.c_int, C.c_double, C.POINTER(PC)]
    L.orc_farneback_prepare_poly(5, 1.1, C.byref(pc))
    ref = NR.farneback_prepare_poly(5, 1.1)
    assert np.allclose(np.array(pc.g[:6]), ref["g"], rtol=0, atol=0)
    assert np.allclose([pc.ig11, pc.ig03, pc.ig33, pc.ig55], [ref["ig11"], ref["ig03"], ref["ig33"], ref["ig55"]],
                       rtol=1e-6)
    k = np.zeros(3, np.float32)
    L.orc_farneback_gaussian_kernel.argtypes = [C.c_int, C.c_double, np.ctypeslib.ndpointer(np.float32)]
    L.orc_farneback_gaussian_kernel(3, 0.0, k)
    assert list(k) == [0.25, 0.5, 0.25]


def test_golden_vectors(oracle):
    g = np.load(os.path.join(GOLDEN, "farneback_golden.npz"))
    for key in [k[:-5] for k in g.files if k.endswith("_flow")]:
        w, h, seed, t0, t1 = [int(v) for v in g[key + "_meta"]]
        clip = SynthClip(w, h, seed)
        flow = oracle.farneback_calc(clip.frame(t0), clip.frame(t1))
        assert np.array_equal(flow, g[key + "_flow"]), key
