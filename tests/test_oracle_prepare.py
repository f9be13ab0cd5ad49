"""CPU checks of the frame-preparation oracle (oracle/prepare_oracle.c: cvtColor BGR2GRAY + cv::resize as
DenseFlow::load_frames_batch calls them, reference src/denseflow_gpu.cpp:163, :169).  Parity unpinned (OpenCV
is not available): known answers, an independent NumPy restatement, frozen goldens."""
import os

import numpy as np
import pytest

from tests import numpy_restatement as NR
from tests.golden.make_prepare_golden import CASES, source

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "prepare_golden.npz")

SIZES = [(64, 48, 32, 24), (64, 48, 40, 30), (100, 70, 224, 224), (340, 256, 224, 224), (33, 17, 64, 64),
         (640, 360, 455, 256), (7, 5, 3, 2), (5, 7, 11, 13), (64, 48, 64, 48), (2, 2, 1, 1), (1, 1, 5, 4),
         (31, 9, 30, 9), (300, 200, 301, 199)]


@pytest.mark.parametrize("sw,sh,dw,dh", SIZES)
@pytest.mark.parametrize("ch", [1, 3])
def test_oracle_equals_numpy_restatement(oracle, sw, sh, dw, dh, ch):
    src = np.random.default_rng(sw * 7 + dh).integers(0, 256, (sh, sw) if ch == 1 else (sh, sw, 3), dtype=np.uint8)
    assert np.array_equal(oracle.prepare_frame(src, dw, dh), NR.prepare_frame(src, dw, dh))


def test_gray_weights_known_answers(oracle):
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 90]]], np.uint8)
    g = oracle.prepare_frame(px, 6, 1)[0].tolist()
    # B=255 -> (255*3735 + 16384) >> 15 = 29 ; G -> 150 ; R -> 76 ; mixed: (10*3735 + 200*19235 + 90*9798 + 16384) >> 15
    assert g == [255, 0, 29, 150, 76, (10 * 3735 + 200 * 19235 + 90 * 9798 + 16384) >> 15]


@pytest.mark.parametrize("dw,dh", [(10, 7), (64, 48), (100, 3), (31, 57), (32, 24)])
def test_constant_image_stays_constant(oracle, dw, dh):
    for v in (0, 1, 127, 254, 255):
        src = np.full((48, 64), v, np.uint8)
        assert np.all(oracle.prepare_frame(src, dw, dh) == v), (v, dw, dh)


def test_same_size_is_a_copy_and_exact_half_is_the_2x2_mean(oracle):
    src = np.random.default_rng(3).integers(0, 256, (48, 64), dtype=np.uint8)
    assert np.array_equal(oracle.prepare_frame(src, 64, 48), src)
    s = src.astype(int)
    mean = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(oracle.prepare_frame(src, 32, 24), mean.astype(np.uint8))


def test_close_to_ideal_bilinear_on_a_ramp(oracle):
    sw, sh, dw, dh = 200, 120, 77, 53
    yy, xx = np.mgrid[0:sh, 0:sw]
    src = np.clip(xx * 0.9 + yy * 0.6, 0, 255).astype(np.uint8)
    out = oracle.prepare_frame(src, dw, dh).astype(float)
    fx = np.clip((np.arange(dw) + 0.5) * sw / dw - 0.5, 0, sw - 1)
    fy = np.clip((np.arange(dh) + 0.5) * sh / dh - 0.5, 0, sh - 1)
    ideal = fx[None, :] * 0.9 + fy[:, None] * 0.6
    assert np.abs(out - ideal).max() <= 1.6  # source quantisation (<1) + fixed-point rounding
    assert np.all(np.diff(out, axis=1) >= 0) and np.all(np.diff(out, axis=0) >= 0)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_golden_vectors(oracle, case):
    name, sw, sh, ch, dw, dh = case
    g = np.load(GOLDEN)
    assert np.array_equal(source(sw, sh, ch, 40 + CASES.index(case)), g[name + "_src"])
    assert np.array_equal(oracle.prepare_frame(g[name + "_src"], dw, dh), g[name + "_dst"])
    assert np.array_equal(NR.prepare_frame(g[name + "_src"], dw, dh), g[name + "_dst"])
