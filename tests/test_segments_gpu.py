"""Several short clips in one FlowBuffer (dfx_next_segments, include/dfx.h): the joined call must return, for every clip,
exactly what that clip returns on its own — pairs are formed inside each clip by the reference's rule
(/root/reference/src/denseflow_gpu.cpp:307-316) and never across a boundary — for every output kind and entry point,
with device batches that span clip boundaries, clips shorter than |step| (no pairs), both signs of the step."""
import ctypes as C

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

pytestmark = pytest.mark.gpu

W, H = 64, 48
LENGTHS = [5, 1, 7, 2, 4]


def _clips():
    return [SynthClip(W, H, 100 + i).frames(n) for i, n in enumerate(LENGTHS)]


@pytest.mark.parametrize("algo", ["tvl1", "farn", "brox"])
@pytest.mark.parametrize("step,batch", [(1, 0), (1, 3), (2, 4), (-1, 3), (-2, 0)])
def test_joined_clips_give_each_clips_own_flows(dfx, algo, step, batch):
    clips = _clips()
    joined = [f for c in clips for f in c]
    knobs = {"max_batch": batch} if batch else {}
    with dfx.FlowEngine(W, H, algo, **knobs) as eng:
        alone = [fl for c in clips for fl in eng.calc_optflows(c, step)]
        eng.next_segments(LENGTHS)
        got = eng.calc_optflows(joined, step)
        again = eng.calc_optflows(clips[0], step)  # the declaration applied to one call only
    assert len(got) == len(alone) == sum(max(n - abs(step), 0) for n in LENGTHS)
    for i, (a, b) in enumerate(zip(got, alone)):
        assert np.array_equal(a, b), f"{algo} step {step} batch {batch}: flow {i} differs"
    assert len(again) == max(LENGTHS[0] - abs(step), 0)
    for a, b in zip(again, alone):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("batch", [0, 3])
def test_every_output_kind_and_the_submit_form(dfx, batch):
    clips = _clips()
    joined = [f for c in clips for f in c]
    knobs = {"max_batch": batch} if batch else {}
    with dfx.FlowEngine(W, H, "farn", **knobs) as eng:
        ux, uy, jx, jy = [], [], [], []
        for c in clips:
            a, b = eng.calc_optflows_u8(c, 1, 20)
            ux += a
            uy += b
            a, b = eng.calc_optflows_jpeg(c, 1, 20)
            jx += a
            jy += b
        eng.next_segments(LENGTHS)
        gx, gy = eng.calc_optflows_u8(joined, 1, 20)
        eng.next_segments(LENGTHS)
        hx, hy = eng.calc_optflows_jpeg(joined, 1, 20)
        eng.next_segments(LENGTHS)
        t1, (sx, sy) = eng.submit_optflows(joined, 1, 20)
        eng.next_segments(LENGTHS)
        t2, flows = eng.submit_optflows(joined, 1)
        eng.wait(t1)
        eng.wait(t2)
        ref = [fl for c in clips for fl in eng.calc_optflows(c, 1)]
    assert len(gx) == len(ux) == len(hx) == len(sx) == len(flows) == len(ref)
    for i in range(len(ux)):
        assert np.array_equal(gx[i], ux[i]) and np.array_equal(gy[i], uy[i])
        assert np.array_equal(sx[i], ux[i]) and np.array_equal(sy[i], uy[i])
        assert hx[i] == jx[i] and hy[i] == jy[i]
        assert np.array_equal(flows[i], ref[i])


def test_device_resident_frames(dfx):
    """dfx_calc_batch_device with joined clips; device memory through the library's own allocator (no torch: which HIP
    runtime a process initialises first depends on the order of the tests before this one)."""
    L = dfx.load_library()
    clips = _clips()
    joined = np.ascontiguousarray(np.stack([f for c in clips for f in c]))
    m = sum(max(n - 1, 0) for n in LENGTHS)
    with dfx.FlowEngine(W, H, "tvl1", max_batch=4) as eng:
        ref = [fl for c in clips for fl in eng.calc_optflows(c, 1)]
        d_frames, d_flows = C.c_void_p(), C.c_void_p()
        got = np.zeros((m, H, W, 2), np.float32)
        assert L.dfx_device_malloc(eng._h, C.byref(d_frames), joined.nbytes) == 0
        assert L.dfx_device_malloc(eng._h, C.byref(d_flows), got.nbytes) == 0
        try:
            assert L.dfx_memcpy_h2d(eng._h, d_frames, joined.ctypes.data, joined.nbytes) == 0
            eng.next_segments(LENGTHS)
            eng.calc_optflows_device(d_frames.value, W, W * H, joined.shape[0], 1, d_flows.value, W * H * 2)
            assert L.dfx_memcpy_d2h(eng._h, got.ctypes.data, d_flows, got.nbytes) == 0
        finally:
            L.dfx_device_free(eng._h, d_frames)
            L.dfx_device_free(eng._h, d_flows)
    for i in range(m):
        assert np.array_equal(got[i], ref[i])


def test_a_wrong_declaration_is_an_error_and_is_consumed(dfx):
    L = dfx.load_library()
    frames = SynthClip(W, H, 3).frames(6)
    with dfx.FlowEngine(W, H, "farn") as eng:
        want = eng.calc_optflows(frames, 1)
        fp = (C.c_void_p * 6)(*[f.ctypes.data for f in frames])
        out = [np.zeros((H, W, 2), np.float32) for _ in range(5)]
        op = (C.c_void_p * 5)(*[o.ctypes.data for o in out])
        assert L.dfx_next_segments(eng._h, (C.c_int * 2)(3, 2), 2) == 0  # 5 frames declared, 6 handed over
        assert L.dfx_calc_batch(eng._h, fp, W, 6, 1, op, W * 8) == 1
        assert b"do not add up" in L.dfx_last_error(eng._h)
        assert L.dfx_calc_batch(eng._h, fp, W, 6, 1, op, W * 8) == 0  # the failed call consumed the declaration
        for a, b in zip(out, want):
            assert np.array_equal(a, b)
        assert L.dfx_next_segments(eng._h, (C.c_int * 1)(-1), 1) == 1
        assert L.dfx_next_segments(eng._h, (C.c_int * 2)(3, 3), 2) == 0
        assert L.dfx_next_segments(eng._h, None, 0) == 0  # cancelled
        assert L.dfx_calc_batch(eng._h, fp, W, 6, 1, op, W * 8) == 0
        for a, b in zip(out, want):
            assert np.array_equal(a, b)


def test_batches_of_many_short_clips_at_224(dfx, oracle):
    """The BASELINE configs[3] shape in small: 12 clips of 224 x 224, joined; flows against the oracle for two of them."""
    n_clips, nf = 12, 9
    clips = [SynthClip(224, 224, 1000 + i).frames(nf) for i in range(n_clips)]
    joined = [f for c in clips for f in c]
    with dfx.FlowEngine(224, 224, "tvl1") as eng:
        eng.next_segments([nf] * n_clips)
        got = eng.calc_optflows(joined, 1)
        st = eng.stats()
    assert len(got) == n_clips * (nf - 1)
    assert st.batch >= n_clips * (nf - 1)  # one device batch holds them all
    for ci in (0, n_clips - 1):
        for i in (0, nf - 2):
            want = oracle.tvl1_calc(clips[ci][i], clips[ci][i + 1])
            assert np.array_equal(got[ci * (nf - 1) + i], want), (ci, i)
