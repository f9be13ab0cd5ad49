"""GPU parity tests for -a=farn: the HIP path (through the C ABI) against the CPU oracle, the
committed golden vectors and size-independent properties at BASELINE.json's full size.
Tolerance from BASELINE.json north_star: <= 1e-3 max-abs on u/v before bounding.  The device runs
the oracle's arithmetic in the oracle's order, so an exact-equality test is included as well."""
import os

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

pytestmark = pytest.mark.gpu
TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("w,h,seed,dt", [(64, 48, 3, 1), (97, 61, 9, 1), (224, 224, 1, 1), (300, 200, 6, 2),
                                         (33, 40, 2, 1), (640, 360, 4, 1)])
def test_single_pair_matches_oracle(dfx, oracle, w, h, seed, dt):
    clip = SynthClip(w, h, seed)
    f0, f1 = clip.frame(0), clip.frame(dt)
    ref = oracle.farneback_calc(f0, f1)
    with dfx.FlowEngine(w, h, "farn") as eng:
        out = eng.calc(f0, f1)
    assert np.max(np.abs(out - ref)) <= TOL


@pytest.mark.parametrize("w,h,seed", [(224, 224, 1), (130, 97, 5)])
def test_bit_exact_with_oracle(dfx, oracle, w, h, seed):
    clip = SynthClip(w, h, seed)
    f0, f1 = clip.frame(0), clip.frame(1)
    ref = oracle.farneback_calc(f0, f1)
    with dfx.FlowEngine(w, h, "farn") as eng:
        out = eng.calc(f0, f1)
    assert np.array_equal(out, ref), f"max-abs {np.max(np.abs(out - ref))}"


def test_golden_vectors(dfx):
    g = np.load(os.path.join(GOLDEN, "farneback_golden.npz"))
    for key in [k[:-5] for k in g.files if k.endswith("_flow")]:
        w, h, seed, t0, t1 = [int(v) for v in g[key + "_meta"]]
        clip = SynthClip(w, h, seed)
        with dfx.FlowEngine(w, h, "farn") as eng:
            out = eng.calc(clip.frame(t0), clip.frame(t1))
        assert np.max(np.abs(out - g[key + "_flow"])) <= TOL, key


@pytest.mark.parametrize("step", [1, -2])
def test_flowbuffer_pair_selection_and_batching(dfx, oracle, step):
    w, h, n = 96, 80, 7
    frames = SynthClip(w, h, 21).frames(n)
    with dfx.FlowEngine(w, h, "farn", max_batch=3) as eng:
        flows = eng.calc_optflows(frames, step)
    m = n - abs(step)
    assert len(flows) == m
    for i in range(m):
        a = i if step > 0 else i - step
        b = i + step if step > 0 else i
        assert np.max(np.abs(flows[i] - oracle.farneback_calc(frames[a], frames[b]))) <= TOL, (step, i)


def test_parameters_and_unsupported_variants(dfx, oracle):
    w, h = 128, 96
    clip = SynthClip(w, h, 13)
    f0, f1 = clip.frame(0), clip.frame(1)
    p = oracle.farneback_default_params()
    p.num_levels, p.win_size, p.num_iters = 2, 9, 4
    ref = oracle.farneback_calc(f0, f1, p)
    with dfx.FlowEngine(w, h, "farn", farn_num_levels=2, farn_win_size=9, farn_num_iters=4) as eng:
        out = eng.calc(f0, f1)
    assert np.max(np.abs(out - ref)) <= TOL
    with pytest.raises(dfx.DfxError) as e:
        dfx.FlowEngine(w, h, "farn", farn_flags=256)  # OPTFLOW_FARNEBACK_GAUSSIAN: denseflow never sets it
    assert e.value.status == 4


def test_full_size_1080p(dfx, oracle):
    w, h = 1920, 1080
    clip = SynthClip(w, h, 2)
    f0, f1 = clip.frame(0), clip.frame(1)
    with dfx.FlowEngine(w, h, "farn") as eng:
        out = eng.calc(f0, f1)
        again = eng.calc(f0, f1)
    assert np.array_equal(out, again)
    ref = oracle.farneback_calc(f0, f1)
    assert np.max(np.abs(out - ref)) <= TOL
    gt = clip.true_flow(0, 1)
    assert np.abs(out - gt)[64:-64, 64:-64].mean() < 0.1


def test_device_resident_entry_point(dfx):
    import torch

    w, h, n = 224, 160, 6
    frames = SynthClip(w, h, 77).frames(n)
    with dfx.FlowEngine(w, h, "farn", max_batch=4) as eng:
        host = eng.calc_optflows(frames, 1)
        d_frames = torch.from_numpy(np.stack(frames)).cuda()
        d_flows = torch.empty((n - 1, h, w, 2), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        eng.calc_optflows_device(d_frames.data_ptr(), w, w * h, n, 1, d_flows.data_ptr(), w * h * 2)
        got = d_flows.cpu().numpy()
    for i in range(n - 1):
        assert np.array_equal(got[i], host[i])


@pytest.mark.parametrize("w,h,seed", [(256, 128, 3), (640, 360, 4), (97, 61, 9), (33, 40, 2), (1920, 1080, 2)])
def test_frame_preparation_variants_do_not_change_a_bit(dfx, oracle, w, h, seed):
    """Two restructurings of the per-frame kernels keep every bit: bilinear pyramid taps whose weight is exactly 0 are
    not evaluated (3 of 4 taps wherever the level's size divides the frame's; DFX_VAR_FARN_EVAL_ZERO_TAPS evaluates
    them), and the polynomial expansion walks 16 rows per workgroup with its vertical window in registers
    (DFX_VAR_FARN_POLY_ONE_ROW: the first form)."""
    from denseflow_amd import engine as E

    clip = SynthClip(w, h, seed)
    frames = clip.frames(4)
    with dfx.FlowEngine(w, h, "farn", max_batch=2, variant=E.VAR_FARN_EVAL_ZERO_TAPS | E.VAR_FARN_POLY_ONE_ROW) as eng:
        base = eng.calc_optflows(frames, 1)
    if w * h <= 640 * 360:
        assert np.array_equal(base[0], oracle.farneback_calc(frames[0], frames[1]))
    for variant in (E.VAR_FARN_POLY_ONE_ROW, E.VAR_FARN_EVAL_ZERO_TAPS, 0):
        with dfx.FlowEngine(w, h, "farn", max_batch=2, variant=variant) as eng:
            out = eng.calc_optflows(frames, 1)
        for i, (a, b) in enumerate(zip(out, base)):
            assert np.array_equal(a, b), f"variant={variant}: pair {i} changed"


@pytest.mark.parametrize("w,h,seed,iters", [(256, 128, 3, 10), (640, 360, 4, 10), (97, 61, 9, 10), (33, 40, 2, 10),
                                            (130, 97, 5, 3), (64, 64, 8, 1), (1920, 1080, 2, 10), (1000, 77, 6, 10),
                                            (70, 500, 12, 4), (129, 49, 14, 10), (65, 43, 15, 2)])
def test_iteration_kernel_forms_do_not_change_a_bit(dfx, oracle, w, h, seed, iters):
    """Round 4: the default iteration kernel recomputes updateMatrices from the previous launch's flow while it streams down
    64-column strips (M never in HBM; the flow ping-pongs between its two plane sets, so an odd iteration count ends in the
    other set).  It must produce the bits of the M-in-HBM kernel of rounds 1-3 (DFX_VAR_FARN_M_IN_HBM), of the simple
    kernels (impl = 1) and of the oracle — for even and odd iteration counts, strips that touch every border, segments
    shorter and longer than a 6-row step, and batches."""
    from denseflow_amd import engine as E

    frames = SynthClip(w, h, seed).frames(4)
    kw = dict(max_batch=2, farn_num_iters=iters)
    with dfx.FlowEngine(w, h, "farn", **kw) as eng:
        out = eng.calc_optflows(frames, 1)
    with dfx.FlowEngine(w, h, "farn", variant=E.VAR_FARN_M_IN_HBM, **kw) as eng:
        in_hbm = eng.calc_optflows(frames, 1)
    for i, (a, b) in enumerate(zip(out, in_hbm)):
        assert np.array_equal(a, b), f"pair {i}: the row-stream kernel differs from the M-in-HBM kernel"
    if w * h <= 640 * 360:
        with dfx.FlowEngine(w, h, "farn", impl=1, **kw) as eng:
            simple = eng.calc_optflows(frames, 1)
        p = oracle.farneback_default_params()
        p.num_iters = iters
        for i in range(3):
            assert np.array_equal(out[i], simple[i]), f"pair {i}: differs from impl = 1"
            assert np.array_equal(out[i], oracle.farneback_calc(frames[i], frames[i + 1], p)), f"pair {i}: differs from the oracle"


@pytest.mark.parametrize("w,h", [(256, 128), (640, 360), (200, 300)])
def test_iteration_kernel_on_unrelated_frames(dfx, oracle, w, h):
    """Frames that have nothing to do with each other (two different textures, and a texture against noise): the flow is
    large and erratic, taps leave the image, and neighbouring pixels no longer sample neighbouring taps — the per-pixel
    gather path (no shared 16-byte window) and the invalid-tap handling of the default iteration kernel run.  Same bits as the
    M-in-HBM kernel, the simple kernels and the oracle."""
    from denseflow_amd import engine as E

    rng = np.random.default_rng(w * 1000 + h)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    frames = [SynthClip(w, h, 31).frame(0), SynthClip(w, h, 32).frame(5), noise, SynthClip(w, h, 31).frame(40)]
    with dfx.FlowEngine(w, h, "farn", max_batch=2) as eng:
        out = eng.calc_optflows(frames, 1)
    with dfx.FlowEngine(w, h, "farn", max_batch=2, variant=E.VAR_FARN_M_IN_HBM) as eng:
        in_hbm = eng.calc_optflows(frames, 1)
    assert max(float(np.abs(f).max()) for f in out) > 8.0, "the case is meant to produce flows that vary by many pixels"
    for i in range(3):
        assert np.array_equal(out[i], in_hbm[i]), f"pair {i}: differs from the M-in-HBM kernel"
        assert np.array_equal(out[i], oracle.farneback_calc(frames[i], frames[i + 1])), f"pair {i}: differs from the oracle"
