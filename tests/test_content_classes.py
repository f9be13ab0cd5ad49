"""CPU twin of tests/test_content_classes_gpu.py: the content classes of real video (denseflow_amd.synth.ContentClip —
letterbox / pillarbox bars, flat frames, clipped plateaus, a fade to black, a hard cut, cartoon step edges, a noisy still)
drawn by hypothesis at <= 64x48 and run through the C oracle (oracle/) and the independent NumPy restatement
(tests/numpy_restatement.py).  Two restatements written apart from each other, from SURVEY.md's appendices, agree bit for
bit on inputs whose gradients, data terms and flows are exactly zero over whole regions — the branches
(`grad <= FLT_EPSILON`, the all-zero png bound rule) the sinusoid clips never take."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from denseflow_amd.synth import CONTENT_CLASSES, ContentClip
from tests import numpy_restatement as NR

_common = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


def _pair(kind, w, h, seed, which):
    clip = ContentClip(w, h, seed, kind)
    pairs = clip.pairs()
    t0, t1 = pairs[which % len(pairs)]
    return clip.frame(t0), clip.frame(t1)


def test_generators_make_what_they_say():
    w, h = 96, 64
    lb, pb = ContentClip(w, h, 1, "letterbox").frame(3), ContentClip(w, h, 1, "pillarbox").frame(3)
    assert (lb[: h // 8] == 0).all() and (lb[-(h // 8):] == 0).all() and lb[h // 8: -(h // 8)].min() > 0
    assert (pb[:, : w // 8] == 0).all() and (pb[:, -(w // 8):] == 0).all() and pb[:, w // 8: -(w // 8)].min() > 0
    c = ContentClip(w, h, 1, "constant")
    assert np.array_equal(c.frame(0), c.frame(4)) and np.unique(c.frame(0)).size == 1
    s = ContentClip(w, h, 1, "constant_step")
    assert np.unique(s.frame(0)).size == 1 and s.frame(0)[0, 0] != s.frame(1)[0, 0]
    sat = ContentClip(w, h, 1, "saturated").frame(0)
    assert (sat == 0).mean() >= 0.2 and (sat == 255).mean() >= 0.2
    f = ContentClip(w, h, 1, "fade")
    assert f.frame(0).max() > 200 and 0 < f.frame(4).max() < 60 and not f.frame(5).any() and not f.frame(6).any()
    cut = ContentClip(w, h, 1, "cut")
    assert np.abs(cut.frame(0).astype(int) - cut.frame(1)).mean() > 20 > np.abs(cut.frame(1).astype(int) - cut.frame(2)).mean()
    car = ContentClip(w, h, 1, "cartoon").frame(0)
    assert np.unique(car).size <= 5 and (car[:, 1:] != car[:, :-1]).mean() < 0.2
    n0, n1 = ContentClip(w, h, 1, "static_noise").frame(0).astype(int), ContentClip(w, h, 1, "static_noise").frame(1).astype(int)
    assert 4.0 < (n1 - n0).std() / np.sqrt(2.0) < 6.0  # 2 % of full scale
    assert np.array_equal(ContentClip(w, h, 1, "static_noise").frame(1), n1.astype(np.uint8))  # seeded


@settings(max_examples=30, **_common)
@given(kind=st.sampled_from(CONTENT_CLASSES), w=st.integers(16, 64), h=st.integers(16, 48), seed=st.integers(0, 10 ** 6),
       which=st.integers(0, 2))
def test_tvl1_oracle_equals_numpy(oracle, kind, w, h, seed, which):
    f0, f1 = _pair(kind, w, h, seed, which)
    flow, tr = oracle.tvl1_calc(f0, f1, want_trace=True)
    mine, iters = NR.tvl1_calc(f0, f1)
    assert np.isfinite(flow).all()
    assert [r[:5] for r in tr.iters_table()] == iters
    # the NumPy form sums the convergence error in another order (double): where that flips a check the test above
    # fails first; otherwise every float32 operation is the same one
    assert np.array_equal(flow, mine), np.max(np.abs(flow - mine))
    if kind in ("constant", "constant_step"):
        assert not flow.any() and all(r[:5] == [2] * 5 for r in tr.iters_table())


@settings(max_examples=40, **_common)
@given(kind=st.sampled_from(CONTENT_CLASSES), w=st.integers(16, 64), h=st.integers(16, 48), seed=st.integers(0, 10 ** 6),
       which=st.integers(0, 2))
def test_farneback_oracle_equals_numpy(oracle, kind, w, h, seed, which):
    f0, f1 = _pair(kind, w, h, seed, which)
    flow = oracle.farneback_calc(f0, f1)
    assert np.isfinite(flow).all()
    assert np.max(np.abs(flow - NR.farneback_calc(f0, f1))) <= 2e-4  # tests/test_oracle_farneback.py's bar
    if kind == "constant":
        assert not flow.any()


@settings(max_examples=8, **_common)
@given(kind=st.sampled_from(CONTENT_CLASSES), w=st.integers(16, 40), h=st.integers(16, 32), seed=st.integers(0, 10 ** 6),
       which=st.integers(0, 2))
def test_brox_oracle_equals_numpy(oracle, kind, w, h, seed, which):
    f0, f1 = _pair(kind, w, h, seed, which)
    flow = oracle.brox_calc(f0, f1)
    assert np.isfinite(flow).all()
    assert np.max(np.abs(flow - NR.brox_calc(f0, f1))) <= 1e-5  # tests/test_oracle_brox.py's bar
    if kind == "constant":
        assert not flow.any()
    elif kind == "constant_step":  # the pyramid's bilinear weights do not sum to exactly 1: gradients of 1e-8 at coarse levels
        assert np.abs(flow).max() < 1e-3


@pytest.mark.parametrize("kind", ["constant", "constant_step", "fade"])
def test_all_zero_flows_through_the_save_stage_oracles(oracle, kind):
    """A flat pair's flow is exactly zero: bounded planes are 128 everywhere (255 * 20 / 40 = 127.5 rounds to even) and
    the png scheme's adaptive bound takes its `0 % 8 == 0 -> += 4` step (src/common.cpp:24-31)."""
    clip = ContentClip(48, 32, 3, kind)
    t0, t1 = clip.pairs()[-1]
    flow = oracle.tvl1_calc(clip.frame(t0), clip.frame(t1))
    assert not flow.any()
    x, y = oracle.flow_to_u8(flow, -20, 20)
    assert (x == 128).all() and (y == 128).all()
    nx, ny = NR.flow_to_u8(flow, -20, 20)
    assert np.array_equal(x, nx) and np.array_equal(y, ny)
    px, py, b, bgr = oracle.flow_to_png_planes(flow)
    assert b == (4.0, 4.0) and (bgr[..., 2] == 1).all()
    if oracle.ref_png_available():  # the reference's own lines (oracle/_ref, built where /root/reference exists)
        assert np.array_equal(oracle.ref_flow_to_png_image(flow), bgr)
