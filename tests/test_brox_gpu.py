"""GPU parity tests for -a=brox (through the C ABI) against the CPU oracle that DEFINES the algorithm
(oracle/brox_oracle.h; reference parity unpinned).  Tolerance 1e-3 max-abs as for
the other algorithms; the device evaluates the oracle's expressions in the same order, so exact equality
is tested too."""
import os

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

pytestmark = pytest.mark.gpu
TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("w,h,seed,dt", [(64, 48, 3, 1), (97, 61, 9, 1), (224, 224, 1, 1), (320, 200, 6, 2),
                                         (20, 33, 2, 1)])
def test_single_pair_matches_oracle(dfx, oracle, w, h, seed, dt):
    clip = SynthClip(w, h, seed)
    f0, f1 = clip.frame(0), clip.frame(dt)
    ref = oracle.brox_calc(f0, f1)
    with dfx.FlowEngine(w, h, "brox") as eng:
        out = eng.calc(f0, f1)
    assert np.max(np.abs(out - ref)) <= TOL


def test_bit_exact_with_oracle(dfx, oracle):
    clip = SynthClip(224, 224, 1)
    f0, f1 = clip.frame(0), clip.frame(1)
    ref = oracle.brox_calc(f0, f1)
    with dfx.FlowEngine(224, 224, "brox") as eng:
        out = eng.calc(f0, f1)
    assert np.array_equal(out, ref), f"max-abs {np.max(np.abs(out - ref))}"


def test_golden_vectors_and_zero_motion(dfx):
    g = np.load(os.path.join(GOLDEN, "brox_golden.npz"))
    for key in [k[:-5] for k in g.files if k.endswith("_flow")]:
        w, h, seed, t0, t1 = [int(v) for v in g[key + "_meta"]]
        clip = SynthClip(w, h, seed)
        with dfx.FlowEngine(w, h, "brox") as eng:
            out = eng.calc(clip.frame(t0), clip.frame(t1))
            zero = eng.calc(clip.frame(t0), clip.frame(t0))
        assert np.max(np.abs(out - g[key + "_flow"])) <= TOL, key
        assert np.all(zero == 0.0)


def test_flowbuffer_step2_batching_and_parameters(dfx, oracle):
    """BASELINE config 5 uses -s=2: flow i is frame i -> i+2."""
    w, h, n = 96, 80, 6
    frames = SynthClip(w, h, 21).frames(n)
    with dfx.FlowEngine(w, h, "brox", max_batch=3) as eng:
        flows = eng.calc_optflows(frames, 2)
    assert len(flows) == n - 2
    for i in range(n - 2):
        assert np.max(np.abs(flows[i] - oracle.brox_calc(frames[i], frames[i + 2]))) <= TOL, i
    p = oracle.brox_default_params()
    p.inner_iterations, p.solver_iterations, p.outer_iterations = 3, 4, 5
    ref = oracle.brox_calc(frames[0], frames[1], p)
    with dfx.FlowEngine(w, h, "brox", brox_inner_iterations=3, brox_solver_iterations=4,
                        brox_outer_iterations=5) as eng:
        out = eng.calc(frames[0], frames[1])
    assert np.max(np.abs(out - ref)) <= TOL


def test_large_pyramid_1080p(dfx, oracle):
    w, h = 1920, 1080
    clip = SynthClip(w, h, 5)
    f0, f1 = clip.frame(0), clip.frame(2)
    with dfx.FlowEngine(w, h, "brox", max_batch=2) as eng:
        out = eng.calc(f0, f1)
        st = eng.stats()
    assert st.levels == len(oracle.brox_pyramid_sizes(w, h)) == 21
    ref = oracle.brox_calc(f0, f1)
    assert np.max(np.abs(out - ref)) <= TOL
    gt = clip.true_flow(0, 2)
    assert np.abs(out - gt)[64:-64, 64:-64].mean() < 0.05


def test_fused_sor_equals_one_launch_per_half_sweep(dfx, oracle):
    """The fused SOR kernel (LDS tile split by column parity, recomputed halo, five sweeps per launch, 8-byte loads, the
    two pixels of a half sweep as packed float2 math, exact Newton reciprocals) must not change a bit relative to the
    simple one-launch-per-half-sweep form, for even and odd solver-iteration counts — as persistent workgroups that
    prefetch the next tile's coefficient planes by LDS-DMA while they sweep (round 6), as one workgroup per tile
    (DFX_VAR_BROX_SOR_PER_TILE, rounds 2-5), and in both synchronisation forms of the latter: a workgroup barrier per half sweep (the default) and band-wise progress counters (round 6,
    DFX_VAR_BROX_SOR_PROGRESS: a wave waits for the two bands next to it, not for the workgroup; measured slower, kept)."""
    from denseflow_amd import engine as E

    # large enough that workgroups of one launch are NOT all co-resident: an in-place update of du/dv would
    # race with neighbours reading their halo (this caught exactly that bug; the kernels ping-pong two sets).
    # odd width and height: the right-most 8-byte pair and the last patch row straddle the image border
    for (w, h, seed) in ((1000, 600, 8), (333, 201, 5)):
        clip = SynthClip(w, h, seed)
        f0, f1 = clip.frame(0), clip.frame(1)
        for solver in (10, 3, 7):
            with dfx.FlowEngine(w, h, "brox", impl=1, brox_solver_iterations=solver) as eng:
                simple = eng.calc(f0, f1)
            for variant in (0, E.VAR_BROX_SOR_PER_TILE, E.VAR_BROX_SOR_PROGRESS):
                with dfx.FlowEngine(w, h, "brox", brox_solver_iterations=solver, variant=variant) as eng:
                    fused = eng.calc(f0, f1)
                    again = eng.calc(f0, f1)  # (a race between bands would not repeat itself)
                assert np.array_equal(simple.view(np.uint32), fused.view(np.uint32)), (w, h, solver, variant)
                assert np.array_equal(fused.view(np.uint32), again.view(np.uint32)), (w, h, solver, variant)


def test_streaming_sor_odd_sizes_and_short_sweep_counts(dfx):
    """The streaming form (k_brox_sor_stream) only runs where a launch has at least four tiles per CU, which the single pairs
    of the test above never reach below 1080p: here six frames of an ODD-sized clip go through one FlowBuffer (5 pairs x 23 x 14
    tiles = 1610 at level 0, 1035 at level 1), so that its special cases are on the path — the last column's 8-byte store whose
    second half lands in the row's padding, tile rows and columns cut by the border, launches of 3 and 2 sweeps (fewer than the
    three sweeps by which the next tile's loads run ahead), the sink stores of lanes that own nothing — and must not change
    a bit relative to one workgroup per tile and to the one-launch-per-half-sweep form."""
    from denseflow_amd import engine as E

    w, h, n = 999, 601, 6
    frames = SynthClip(w, h, 13).frames(n)
    for solver in (10, 3, 7):
        with dfx.FlowEngine(w, h, "brox", impl=1, brox_solver_iterations=solver, brox_outer_iterations=3, max_batch=n) as eng:
            simple = eng.calc_optflows(frames, 1)
        for variant in (0, E.VAR_BROX_SOR_PER_TILE):
            with dfx.FlowEngine(w, h, "brox", brox_solver_iterations=solver, brox_outer_iterations=3, variant=variant,
                                max_batch=n) as eng:
                fused = eng.calc_optflows(frames, 1)
                again = eng.calc_optflows(frames, 1)
            assert len(fused) == n - 1
            for i in range(n - 1):
                assert np.array_equal(simple[i].view(np.uint32), fused[i].view(np.uint32)), (solver, variant, i)
                assert np.array_equal(fused[i].view(np.uint32), again[i].view(np.uint32)), (solver, variant, i)


@pytest.mark.parametrize("w,h,seed,t0,t1", [(224, 224, 1, 0, 8), (80, 56, 21, 0, 2), (40, 33, 4, 0, 3)])
def test_mirror_index_fast_path_and_its_fallback(dfx, oracle, w, h, seed, t0, t1):
    """Round 5: stage 1's four mirror indices take ONE reflection when the bilinear window lies within one image size of
    the frame (mirror_idx_near) and the general modulo form otherwise, and 1 / sqrtf is an exact 16-instruction sequence.
    Large flows (an 8-frame jump), tiny pyramid levels: still the oracle's bits."""
    clip = SynthClip(w, h, seed)
    f0, f1 = clip.frame(t0), clip.frame(t1)
    with dfx.FlowEngine(w, h, "brox", max_batch=2) as eng:
        out = eng.calc(f0, f1)
    assert np.array_equal(out, oracle.brox_calc(f0, f1))


def test_config5_shape_4k_step2(dfx, oracle):
    """BASELINE config 5 at its stated frame size: 3840x2160, -a=brox -s=2 (reference call
    src/denseflow_gpu.cpp:303, :331-334 with the pair rule of :315-316).  Four frames through the FlowBuffer entry
    point give two flows (0 -> 2, 1 -> 3); both are compared with the oracle, and the 24-level pyramid is checked."""
    w, h, n = 3840, 2160, 4
    clip = SynthClip(w, h, 5)  # config 5 is seed 5
    frames = clip.frames(n)
    with dfx.FlowEngine(w, h, "brox") as eng:
        flows = eng.calc_optflows(frames, 2)
        st = eng.stats()
    assert len(flows) == n - 2
    sizes = oracle.brox_pyramid_sizes(w, h)
    assert len(sizes) == 24 and st.levels == 24
    assert [(st.level_w[l], st.level_h[l]) for l in range(24)] == [tuple(sz) for sz in sizes]
    for i in range(n - 2):
        ref = oracle.brox_calc(frames[i], frames[i + 2])
        assert np.max(np.abs(flows[i] - ref)) <= TOL, i
        assert np.array_equal(flows[i], ref), i
    gt = clip.true_flow(0, 2)
    assert np.abs(flows[0] - gt)[128:-128, 128:-128].mean() < 0.05
