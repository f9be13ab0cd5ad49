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
    so = os.path.join(out_dir, "libctrl_harness.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
           os.path.join(HERE, "ctrl_harness.cpp"), os.path.join(ROOT, "oracle", "liboracle.so"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.run(cmd, check=True, capture_output=True)
    L = C.CDLL(so)
    L.ctrl_replay_level.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_double, C.c_double, i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ctrl_replay_level.restype = C.c_int
    L.ctrl_replay_level_ex.argtypes = L.ctrl_replay_level.argtypes + [C.c_int]
    L.ctrl_replay_level_ex.restype = C.c_int
    return L


def _level_inputs(w, h, seed, dt):
    clip = SynthClip(w, h, seed)
    return clip.frame(0).astype(np.float32), clip.frame(dt).astype(np.float32)


@pytest.mark.parametrize("split_warp", [0, 1])
@pytest.mark.parametrize("fuse_k", [1, 2, 3, 4, 8, 16])
@pytest.mark.parametrize("w,h,seed,dt,iterations", [(48, 40, 2, 1, 300), (40, 32, 4, 3, 300), (40, 32, 4, 3, 37),
                                                    (40, 32, 4, 1, 1), (40, 32, 4, 1, 2), (40, 32, 4, 1, 0)])
def test_state_machine_replays_oracle(oracle, harness, fuse_k, w, h, seed, dt, iterations, split_warp):
    I0, I1 = _level_inputs(w, h, seed, dt)
    prm = oracle.tvl1_default_params()
    prm.iterations = iterations
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
