"""Edge-of-range frame sizes for all three algorithms, GPU vs oracle (bit-exact): the smallest frames
(pyramids that collapse to one level, tiles that are mostly halo), extreme aspect ratios (one tile row or
one tile column), sizes one off the tile/vector widths, and BASELINE's largest size (3840x2160, one pair).
"""
import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

pytestmark = pytest.mark.gpu

SMALL = [(16, 16), (17, 16), (20, 17), (31, 33), (63, 9), (65, 8), (8, 64), (1024, 16), (16, 512), (257, 129)]


def _pair(w, h, seed=4):
    rng = np.random.default_rng(seed + w * 31 + h)
    if min(w, h) < 32:  # SynthClip's longest wavelengths are meaningless here: smooth noise with a shift instead
        base = rng.uniform(0, 255, (h + 4, w + 4))
        for _ in range(2):
            base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 0) + np.roll(base, -1, 1)) / 5
        base = (base - base.min()) / max(np.ptp(base), 1e-9) * 255
        f0 = base[2:-2, 2:-2].astype(np.uint8)
        f1 = base[2:-2, 1:-3].astype(np.uint8)
        return np.ascontiguousarray(f0), np.ascontiguousarray(f1)
    clip = SynthClip(w, h, seed)
    return clip.frame(0), clip.frame(1)


@pytest.mark.parametrize("w,h", SMALL)
def test_tvl1_small_and_skewed_frames(dfx, oracle, w, h):
    f0, f1 = _pair(w, h)
    ref, tr = oracle.tvl1_calc(f0, f1, want_trace=True)
    with dfx.FlowEngine(w, h, "tvl1") as eng:
        out = eng.calc(f0, f1)
        st = eng.stats()
    assert st.levels == tr.nscales
    assert np.array_equal(out, ref), f"max-abs {np.max(np.abs(out - ref))}"


@pytest.mark.parametrize("w,h", SMALL)
def test_farneback_small_and_skewed_frames(dfx, oracle, w, h):
    f0, f1 = _pair(w, h)
    ref = oracle.farneback_calc(f0, f1)
    with dfx.FlowEngine(w, h, "farn") as eng:
        out = eng.calc(f0, f1)
    assert np.array_equal(out, ref), f"max-abs {np.max(np.abs(out - ref))}"


@pytest.mark.parametrize("w,h", SMALL)
def test_brox_small_and_skewed_frames(dfx, oracle, w, h):
    f0, f1 = _pair(w, h)
    ref = oracle.brox_calc(f0, f1)
    with dfx.FlowEngine(w, h, "brox") as eng:
        out = eng.calc(f0, f1)
    assert np.all(np.isfinite(out))
    assert np.array_equal(out, ref), f"max-abs {np.max(np.abs(out - ref))}"


def test_tvl1_largest_baseline_size_3840x2160(dfx, oracle):
    """BASELINE config 5's frame size through the TVL1 path: one pair against the oracle (about 10 s of CPU),
    identical executed iteration counts, zero motion exactly zero."""
    w, h = 3840, 2160
    clip = SynthClip(w, h, 5)
    f0, f1 = clip.frame(0), clip.frame(2)
    ref, tr = oracle.tvl1_calc(f0, f1, want_trace=True)
    with dfx.FlowEngine(w, h, "tvl1", max_batch=2) as eng:
        out = eng.calc(f0, f1)
        st = eng.stats()
        zero = eng.calc(f1, f1)
    assert [r[:5] for r in st.iters_table()] == [r[:5] for r in tr.iters_table()]
    assert np.array_equal(out, ref), f"max-abs {np.max(np.abs(out - ref))}"
    assert np.all(zero == 0.0)


def test_farneback_3840x2160(dfx, oracle):
    w, h = 3840, 2160
    clip = SynthClip(w, h, 5)
    f0, f1 = clip.frame(0), clip.frame(2)
    ref = oracle.farneback_calc(f0, f1)
    with dfx.FlowEngine(w, h, "farn", max_batch=2) as eng:
        out = eng.calc(f0, f1)
    assert np.array_equal(out, ref), f"max-abs {np.max(np.abs(out - ref))}"
