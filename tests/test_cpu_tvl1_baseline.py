"""CPU cv::optflow::DualTVL1OpticalFlow restatement (oracle/cpu_tvl1_baseline.c) — the
timing comparator of bench.py's `cpu_baseline`.  It is not a parity oracle; these tests pin its building blocks to
known answers and check that the whole thing is a working TV-L1 (so the time it takes is the time of real work)."""
import ctypes as C

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip


def test_median_blur_is_the_exact_median_with_replicated_border(oracle):
    rng = np.random.default_rng(5)
    for (h, w) in [(7, 9), (16, 16), (5, 31)]:
        a = rng.standard_normal((h, w)).astype(np.float32)
        for k in (3, 5):
            out = np.empty_like(a)
            oracle.lib().cpu_tvl1_median_blur(a, out, w, h, k)
            r = k // 2
            p = np.pad(a, r, mode="edge")
            win = np.stack([p[j:j + h, i:i + w] for j in range(k) for i in range(k)], axis=-1)
            assert np.array_equal(out, np.median(win, axis=-1).astype(np.float32)), (h, w, k)


def test_cubic_coefficients_are_opencvs_a_minus_075_kernel(oracle):
    c = (C.c_float * 4)()
    for x in (0.0, 0.25, 0.5, 31 / 32):
        oracle.lib().cpu_tvl1_cubic_coeffs(x, C.byref(c))
        v = np.array(list(c), dtype=np.float64)
        assert abs(v.sum() - 1.0) < 1e-6
        A = -0.75
        ref1 = ((A + 2) * x - (A + 3)) * x * x + 1
        assert abs(v[1] - ref1) < 1e-6
    oracle.lib().cpu_tvl1_cubic_coeffs(0.0, C.byref(c))
    assert list(c) == [0.0, 1.0, 0.0, 0.0]


def test_remap_identity_and_constant_border(oracle):
    rng = np.random.default_rng(2)
    h, w = 24, 40
    a = rng.uniform(0, 255, (h, w)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty_like(a)
    oracle.lib().cpu_tvl1_remap_cubic(a, w, h, xx.copy(), yy.copy(), out)
    assert np.array_equal(out, a)  # integer coordinates: the centre tap has weight 1, the rest 0
    oracle.lib().cpu_tvl1_remap_cubic(a, w, h, (xx + 100).copy(), yy.copy(), out)
    assert np.all(out == 0)  # BORDER_CONSTANT, value 0
    # coordinates are quantised to 1/32 px: x + 1/64 rounds (half to even) onto x or x + 1/32
    oracle.lib().cpu_tvl1_remap_cubic(a, w, h, (xx + 1 / 128).copy(), yy.copy(), out)
    assert np.array_equal(out[4:-4, 4:-4], a[4:-4, 4:-4])


def test_resize_uses_half_pixel_centres(oracle):
    h, w = 10, 20
    ramp = np.tile(np.arange(w, dtype=np.float32), (h, 1))
    dw, dh = 16, 8
    out = np.empty((dh, dw), np.float32)
    oracle.lib().cpu_tvl1_resize_linear(ramp, w, h, out, dw, dh, 0.8, 0.8)
    expect = np.clip((np.arange(dw) + 0.5) * 1.25 - 0.5, 0, w - 1)
    assert np.allclose(out[3], expect, atol=1e-5)
    const = np.full((h, w), 7.5, np.float32)
    oracle.lib().cpu_tvl1_resize_linear(const, w, h, out, dw, dh, 0.8, 0.8)
    assert np.all(out == 7.5)


def test_zero_motion_gives_zero_flow(oracle):
    f = SynthClip(96, 64, 3).frame(0)
    flow, st = oracle.cpu_tvl1_calc(f, f, want_stats=True)
    assert np.all(flow == 0)
    assert st.inner_iterations == st.outer_iterations  # every warp stops after its first inner iteration
    assert st.nscales == 5 and (st.w[0], st.h[0]) == (96, 64)


def test_recovers_a_translation_and_agrees_roughly_with_the_cuda_semantics_oracle(oracle):
    clip = SynthClip(160, 120, 11)
    f0, f1 = clip.frame(0), clip.frame(1)
    flow = oracle.cpu_tvl1_calc(f0, f1)
    gt = clip.true_flow(0, 1)
    inner = (slice(16, -16), slice(16, -16))
    assert np.abs(flow[inner] - gt[inner]).mean() < 0.15
    cuda_like = oracle.tvl1_calc(f0, f1)
    # CPU and CUDA OpenCV differ by far more than 1e-3: same motion, different numbers
    assert np.abs(flow[inner] - cuda_like[inner]).mean() < 0.2
    assert np.abs(flow - cuda_like).max() > 1e-3


def test_native_build_gives_the_portable_builds_flow(oracle):
    clip = SynthClip(80, 56, 4)
    f0, f1 = clip.frame(0), clip.frame(2)
    a = oracle.cpu_tvl1_calc(f0, f1)
    b = oracle.cpu_tvl1_calc(f0, f1, native=True)
    assert np.max(np.abs(a - b)) < 1e-3  # -march=native may vectorise the float error sum differently
