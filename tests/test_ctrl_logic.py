"""CPU test of the device-side TVL1 control state machine (denseflow_amd/csrc/tvl1_ctrl.h).

The header that is compiled into the HIP kernels is compiled here with g++ into a small harness that
uses the oracle's stage functions as stand-ins for the kernels.  For every fuse_k the state machine
must execute exactly the oracle's iterations (same counts, same checks, bit-identical flow)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def harness(oracle):
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libctrl_harness.%d.so" % os.getpid())  # per process: pytest-xdist workers build in parallel
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
           os.path.join(HERE, "ctrl_harness.cpp"), os.path.join(ROOT, "oracle", "liboracle.so"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.run(cmd, check=True, capture_output=True)
    L = C.CDLL(so)
    os.unlink(so)  # the mapping stays valid
    L.ctrl_replay_level.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_double, C.c_double, i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ctrl_replay_level.restype = C.c_int
    L.ctrl_replay_level_ex.argtypes = L.ctrl_replay_level.argtypes + [C.c_int]
    L.ctrl_replay_level_ex.restype = C.c_int
    return L


def _level_inputs(w, h, seed, dt):
    clip = SynthClip(w, h, seed)
    return clip.frame(0).astype(np.float32), clip.frame(dt).astype(np.float32)


@pytest.mark.parametrize("split_warp", [0, 1, 2])
@pytest.mark.parametrize("fuse_k", [1, 2, 3, 4, 8, 16])
@pytest.mark.parametrize("w,h,seed,dt,iterations", [(48, 40, 2, 1, 300), (40, 32, 4, 3, 300), (40, 32, 4, 3, 37),
                                                    (40, 32, 4, 1, 1), (40, 32, 4, 1, 2), (40, 32, 4, 1, 0)])
@pytest.mark.parametrize("epsilon", [0.01, 0.0])
def test_state_machine_replays_oracle(oracle, harness, fuse_k, w, h, seed, dt, iterations, split_warp, epsilon):
    if epsilon == 0.0 and (iterations > 37 or fuse_k not in (1, 4)):
        pytest.skip("epsilon = 0 (no early exit: the head does not end a segment) is replayed on the short loops")
    I0, I1 = _level_inputs(w, h, seed, dt)
    prm = oracle.tvl1_default_params()
    prm.iterations = iterations
    prm.epsilon = epsilon
    # oracle: plain host loop
    u1 = np.zeros((h, w), np.float32)
    u2 = np.zeros((h, w), np.float32)
    tr = oracle.Tvl1Trace()
    oracle.lib().orc_tvl1_proc_one_scale(I0, I1, u1, u2, w, h, C.byref(prm), 0, C.byref(tr))
    # state machine
    v1 = np.zeros((h, w), np.float32)
    v2 = np.zeros((h, w), np.float32)
    iters = np.zeros(16, np.int32)
    nchk, steps = C.c_int(0), C.c_int(0)
    rc = harness.ctrl_replay_level_ex(I0, I1, v1, v2, w, h, prm.warps, prm.iterations, fuse_k, prm.epsilon,
                                      prm.lambda_, prm.theta, prm.tau, iters, C.byref(nchk), C.byref(steps), split_warp)
    assert rc == 0
    assert [int(v) for v in iters[:5]] == [tr.iters[0][k] for k in range(5)]
    assert nchk.value == tr.n_checks
    assert np.array_equal(u1, v1) and np.array_equal(u2, v2)
    # every warp costs one step, every segment ceil(len/fuse_k) steps: never more steps than iterations + warps
    assert steps.value <= int(iters[:5].sum()) + prm.warps
    if split_warp and iterations > 0:  # a warp no longer occupies a step of its own
        assert steps.value <= int(iters[:5].sum())
    if split_warp == 2 and iterations >= 2 and fuse_k >= 2:
        # the head takes the first two iterations with it: a warp whose loop ends at its first check costs one step slot
        # (the launch pair of that step id), not more
        assert steps.value <= prm.warps + sum(max(0, int(n) - 2 + fuse_k - 1) // fuse_k for n in iters[:5])


@pytest.mark.parametrize("shift", [0, 1])
def test_step_tile_geometry_partitions_the_image(harness, shift):
    """The step kernel's tile geometries (tvl1_ctrl.h: the default with tile columns from x = 0, and the classic one):
    every pixel has exactly one owning tile, owned pixels never depend on values outside their tile, and the launcher
    provides exactly the step's tiles — for the 1080p pyramid, degenerate sizes and random ones."""
    harness.ctrl_geometry_check.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)] * 2
    harness.ctrl_geometry_check.restype = C.c_int
    rng = np.random.default_rng(7)
    sizes = [(1920, 1080), (1536, 864), (1229, 691), (983, 553), (786, 442), (224, 224), (16, 16), (57, 40), (64, 64),
             (56, 24), (60, 28), (65, 33), (1, 1), (63, 5), (128, 31), (120, 442)]
    sizes += [(int(rng.integers(1, 400)), int(rng.integers(1, 200))) for _ in range(40)]
    nt, grid = C.c_int(0), C.c_int(0)
    for (w, h) in sizes:
        for k in (1, 2, 3, 4, 6, 12):
            if w * h > 500_000 and k != 4:
                continue
            for split in (0, 1):
                rc = harness.ctrl_geometry_check(w, h, 64, 32, k, shift, split, C.byref(nt), C.byref(grid))
                assert rc == 0 and nt.value == grid.value, (w, h, k, shift, split, rc, nt.value, grid.value)
    # what the default buys at the coarsest 1080p level (K = 4): 14 instead of 15 tile columns
    assert harness.ctrl_geometry_check(786, 442, 64, 32, 4, shift, 1, C.byref(nt), C.byref(grid)) == 0
    assert nt.value == {0: 285, 1: 266}[shift]


def test_xcd_aware_tile_mapping_is_a_bijection(harness):
    """dfx_xcd_tile_index: every tile of a grid is visited exactly once for every tile count (a kernel that skipped or
    doubled a tile would still produce plausible output elsewhere), and each dispatch class gets a contiguous run."""
    harness.ctrl_xcd_map_check.argtypes = [C.c_int]
    harness.ctrl_xcd_map_check.restype = C.c_int
    for nt in list(range(1, 300)) + [1020, 1100, 1248, 1575, 2040, 30 * 270, 65537]:
        assert harness.ctrl_xcd_map_check(nt) == 0, nt


def test_step_work_counts_what_the_trapezoid_layout_executes(harness):
    """tvl1_step_work(n, K) — the half rows a step of n iterations updates on a 32-row tile with a K-row halo, which the
    state machine adds up per level for dfx_stats.tvl1_lane_iters / bench.py's useful_frac — against a row-by-row count of
    tvl1_tile.h's skip rule (with d iterations to go, primal updates are skipped on rows closer than K - d - 1 to the
    tile's top / bottom edge, dual updates on rows closer than K - d)."""
    harness.ctrl_step_work.argtypes = [C.c_int, C.c_int]
    harness.ctrl_step_work.restype = C.c_int
    for K in range(1, 13):
        for n in range(1, K + 1):
            work = 0
            for it in range(n):
                need = K - (n - 1 - it)
                for a in range(16):  # float2 a rows from the edge = two tile rows
                    work += 2 * (0 if a < need - 1 else 1) + 2 * (0 if a < need else 1)
            assert harness.ctrl_step_work(n, K) == work, (n, K)
    assert harness.ctrl_step_work(4, 4) == 224  # of 256: the 12.5 % a full step skips
