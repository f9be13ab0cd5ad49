"""CPU checks of the flow-bounding oracle (oracle/quant_oracle.c = reference src/common.cpp:4-16).

Pinned: the golden file was produced by the reference's own source lines (tests/golden/
make_quant_golden.py), and when oracle/_ref/libref_quant.so is present the oracle is compared with it
directly on fresh inputs as well.
"""
import os

import numpy as np
import pytest

from tests import numpy_restatement as NR
from tests.golden.make_quant_golden import CASES, adversarial_flow

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "quant_golden.npz")


def test_known_answers(oracle):
    f = np.zeros((1, 8, 2), np.float32)
    f[0, :, 0] = [0.0, 20.0, -20.0, 20.000002, -20.000002, np.nan, np.inf, -np.inf]
    f[0, :, 1] = [10.0, -10.0, 0.0392157, -0.0392157, 19.96, -19.96, 1e-30, 5.0]
    x, y = oracle.flow_to_u8(f, -20, 20)
    # 255*(0+20)/40 = 127.5 -> 128 (ties to even); the bounds map to 255 / 0; beyond them the clamps take over
    assert x[0].tolist() == [128, 255, 0, 255, 0, 0, 255, 0]
    # 255*30/40 = 191.25 -> 191 ; 255*10/40 = 63.75 -> 64 ; 127.75 -> 128 ; 127.25 -> 127 ; ... ; 159.375 -> 159
    assert y[0].tolist() == [191, 64, 128, 127, 255, 0, 128, 159]


def test_ties_round_to_even(oracle):
    # bound 32: 255*(v+32)/64 hits k + 0.5 exactly when v = (2k+1)*32/255 - 32 is representable; use bound = 127.5:
    # 255*(v+127.5)/255 = v + 127.5, so integers v give exact ties
    v = np.arange(-127, 128, dtype=np.float32)
    f = np.stack([v, v], -1)[None]
    x, _ = oracle.flow_to_u8(f, -127.5, 127.5)
    expect = np.rint(v.astype(np.float64) + 127.5).astype(np.uint8)  # numpy rint is half-to-even too
    assert np.array_equal(x[0], expect)
    assert x[0][0] == 0 and x[0][1] == 2 and x[0][2] == 2  # 0.5 -> 0, 1.5 -> 2, 2.5 -> 2


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_equals_reference_golden(oracle, case):
    name = case[0]
    g = np.load(GOLDEN)
    bound = float(g[name + "_bound"][0])
    x, y = oracle.flow_to_u8(g[name + "_flow"], -bound, bound)
    assert np.array_equal(x, g[name + "_x"]) and np.array_equal(y, g[name + "_y"])


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_numpy_restatement_equals_reference_golden(case):
    name = case[0]
    g = np.load(GOLDEN)
    bound = float(g[name + "_bound"][0])
    x, y = NR.flow_to_u8(g[name + "_flow"], -bound, bound)
    assert np.array_equal(x, g[name + "_x"]) and np.array_equal(y, g[name + "_y"])


def test_golden_inputs_are_reproducible():
    g = np.load(GOLDEN)
    for name, bound, w, h, seed in CASES:
        assert np.array_equal(adversarial_flow(bound, w, h, seed), g[name + "_flow"], equal_nan=True)


def test_oracle_equals_compiled_reference_on_fresh_inputs(oracle):
    if not oracle.ref_quant_available():
        pytest.skip("oracle/_ref/libref_quant.so not built (needs /root/reference: make -C oracle ref)")
    for seed, (lo, hi) in enumerate([(-20, 20), (-32, 32), (-1, 1), (-5, 20), (0, 0), (3, -3)]):
        flow = adversarial_flow(max(abs(hi), 1), 80, 50, 100 + seed)
        ox, oy = oracle.flow_to_u8(flow, lo, hi)
        rx, ry = oracle.ref_flow_to_u8(flow, lo, hi)
        assert np.array_equal(ox, rx) and np.array_equal(oy, ry), (lo, hi)
        nx, ny = NR.flow_to_u8(flow, lo, hi)
        assert np.array_equal(nx, rx) and np.array_equal(ny, ry), (lo, hi)


# ---- the -st=png scheme: convertFlowToPngImage, /root/reference/src/common.cpp:18-46 --------------------------------
from tests.golden.make_png_planes_golden import CASES as PNG_CASES, flow_with_extrema  # noqa: E402

PNG_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "png_planes_golden.npz")


@pytest.mark.parametrize("case", PNG_CASES, ids=[c[0] for c in PNG_CASES])
def test_png_scheme_oracle_equals_reference_golden(oracle, case):
    """Pinned like the bounding above: the golden images were produced by the reference's own lines
    (tests/golden/make_png_planes_golden.py: oracle/_ref/libref_png.so)."""
    name = case[0]
    g = np.load(PNG_GOLDEN)
    x, y, (bx, by), bgr = oracle.flow_to_png_planes(g[name + "_flow"])
    want = g[name + "_bgr"]
    assert np.array_equal(bgr, want)
    assert np.array_equal(x, want[..., 0]) and np.array_equal(y, want[..., 1])
    h = want.shape[0]
    assert int(want[0, 0, 2]) == min(255, int(np.rint(bx / 4))) and int(want[h - 1, 0, 2]) == min(255, int(np.rint(by / 4)))


def test_png_scheme_bound_rule_known_answers(oracle):
    """ceil((min(extent, max|v|) * 128 / 127) / 4) * 4, capped at 1020, + 4 when a multiple of 8."""
    def bounds(w, h, mu, mv):
        f = np.zeros((h, w, 2), np.float32)
        f[0, 0, 0], f[h - 1, w - 1, 1] = -mu, mv
        return oracle.flow_to_png_planes(f)[2]

    assert bounds(64, 48, 0.0, 0.0) == (4.0, 4.0)          # 0 is a multiple of 8
    assert bounds(64, 48, 3.2, 7.9) == (4.0, 12.0)          # 8 -> 12
    assert bounds(64, 48, 15.5, 11.9) == (20.0, 12.0)       # 16 -> 20
    assert bounds(40, 30, 500.0, 77.0) == (44.0, 36.0)      # min(w, .) = 40 -> 40.31 -> 44; min(h, .) = 30 -> 32 -> 36
    assert bounds(2000, 2000, 1500.0, 1012.0) == (1020.0, 1020.0)  # the 255 * 4 cap (1020 % 8 == 4)


def test_png_scheme_golden_inputs_are_reproducible():
    g = np.load(PNG_GOLDEN)
    for name, w, h, mu, mv, seed, neg in PNG_CASES:
        assert np.array_equal(flow_with_extrema(w, h, mu, mv, seed, neg), g[name + "_flow"])


def test_png_scheme_oracle_equals_compiled_reference_on_fresh_inputs(oracle):
    if not oracle.ref_png_available():
        pytest.skip("oracle/_ref/libref_png.so not built (needs /root/reference: make -C oracle ref)")
    rng = np.random.default_rng(3)
    for k in range(12):
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        flow = flow_with_extrema(w, h, float(rng.choice([0.0, 0.3, 7.9, 15.5, 31.7, 300.0])),
                                 float(rng.choice([0.0, 1.0, 3.9, 23.6, 64.0])), 200 + k, bool(k & 1))
        assert np.array_equal(oracle.flow_to_png_planes(flow)[3], oracle.ref_flow_to_png_image(flow)), (w, h, k)


def _nan_flows(w=33, h=21):
    """minMaxLoc on flows with NaNs: at the very first element (a search seeded with it would be stuck), a whole plane of
    them (nothing located: minMaxLoc reports 0 / 0, the bound rule gives 4 — ADVICE r5), both planes, and a clean one."""
    rng = np.random.default_rng(5)
    base = (rng.standard_normal((h, w, 2)) * np.array([9.0, 3.0])).astype(np.float32)
    first = base.copy(); first[0, 0, :] = np.nan
    u_nan = base.copy(); u_nan[..., 0] = np.nan
    both = np.full_like(base, np.nan)
    sprinkled = base.copy(); sprinkled[rng.random((h, w)) < 0.3] = np.nan
    return {"clean": base, "NaN first": first, "u all NaN": u_nan, "all NaN": both, "30 % NaN": sprinkled}


def test_png_scheme_nan_rules(oracle):
    flows = _nan_flows()
    want = oracle.flow_to_png_planes(flows["clean"])[2]
    assert oracle.flow_to_png_planes(flows["NaN first"])[2] == want  # one NaN pixel does not move the extrema of 693
    x, y, b, _ = oracle.flow_to_png_planes(flows["u all NaN"])
    assert b == (4.0, want[1]) and not x.any()  # NaN -> cvRound = INT_MIN -> saturates to 0
    assert oracle.flow_to_png_planes(flows["all NaN"])[2] == (4.0, 4.0)
    if oracle.ref_png_available():  # the reference's own lines over the stand-in minMaxLoc / convertTo
        for name, f in flows.items():
            assert np.array_equal(oracle.ref_flow_to_png_image(f), oracle.flow_to_png_planes(f)[3]), name
