"""Device JPEG encoder (denseflow_amd/csrc/jpeg_kernels.hip, dfx_calc_batch_jpeg): the files must be, byte for byte,
what libjpeg-turbo writes for the same bounded planes — the library behind the reference's imencode(".jpg"),
/root/reference/src/common.cpp:56-57; live through Pillow, and tests/golden/jpeg_golden.npz — and what the shell's host
encoder writes (src/image_io.cpp: imencodeJpeg, itself pinned to libjpeg in tests/test_jpeg_libjpeg_pin.py).  Integer /
byte work: bit-exact."""
import ctypes as C
import io

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip
from tests.test_host_shell import built, harness  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _libjpeg_file(plane, quality):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(plane), "L").save(b, "JPEG", quality=quality)
    return b.getvalue()


def _host_file(harness, plane, quality):
    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    h, w = plane.shape
    buf = np.zeros(w * h * 3 + 4096, np.uint8)
    plane = np.ascontiguousarray(plane)
    n = harness.hh_encode_jpeg(plane.ctypes.data, w, h, quality, buf.ctypes.data, buf.size)
    assert n > 0
    return buf[:n].tobytes()


@pytest.mark.parametrize("algo,w,h,n,quality,batch", [("farn", 640, 360, 6, 95, 2), ("farn", 70, 45, 4, 95, 0),
                                                      ("tvl1", 224, 224, 5, 95, 3), ("farn", 257, 131, 4, 50, 0),
                                                      ("farn", 64, 64, 3, 100, 0), ("brox", 352, 288, 3, 95, 0),
                                                      ("farn", 1920, 1080, 4, 95, 2), ("farn", 33, 17, 3, 10, 0)])
def test_files_are_libjpegs_and_the_host_encoders_byte_for_byte(dfx, harness, algo, w, h, n, quality, batch):
    from PIL import Image

    frames = SynthClip(w, h, 31).frames(n)
    knobs = {"max_batch": batch} if batch else {}
    with dfx.FlowEngine(w, h, algo, **knobs) as eng:
        px, py = eng.calc_optflows_u8(frames, 1, 20)
        jx, jy = eng.calc_optflows_jpeg(frames, 1, 20, quality)
        jx2, jy2 = eng.calc_optflows_jpeg(frames, 1, 20, quality)  # buffers are reused across calls
    assert len(jx) == len(px) == n - 1
    for i in range(n - 1):
        for plane, got, again in ((px[i], jx[i], jx2[i]), (py[i], jy[i], jy2[i])):
            want = _host_file(harness, plane, quality)
            assert got == want, f"{algo} {w}x{h} q{quality} flow {i}: {len(got)} vs {len(want)} bytes"
            assert again == want
            assert got == _libjpeg_file(plane, quality), f"{algo} {w}x{h} q{quality} flow {i}: not libjpeg's bytes"
            dec = np.array(Image.open(io.BytesIO(got)))
            assert dec.shape == (h, w)
            if quality >= 95:
                assert np.abs(dec.astype(int) - plane.astype(int)).mean() < 1.5


def test_arbitrary_planes_including_stuffing_and_long_zero_runs(dfx, harness):
    """Planes that are NOT flows, pushed through the same kernels (frames whose flow saturates the bound give planes of
    0 / 255 runs; noise frames give busy spectra): still the host encoder's bytes.  A batch that does not compress below
    the 4 bits per pixel the stream buffer is first sized for is coded again after the buffer has grown to what the scan
    pass measured (round 4) — it used to fail with DFX_ERR_UNSUPPORTED and cost the host shell a second flow computation."""
    rng = np.random.default_rng(8)
    w, h = 200, 120
    frames = [rng.integers(0, 256, (h, w)).astype(np.uint8) for _ in range(4)]  # noise: Farneback flows are garbage
    with dfx.FlowEngine(w, h, "farn") as eng:
        px, py = eng.calc_optflows_u8(frames, 1, 0.05)  # tiny bound: almost everything saturates to 0 / 255
        jx, jy = eng.calc_optflows_jpeg(frames, 1, 0.05, 95)
    for i in range(3):
        assert jx[i] == _host_file(harness, px[i], 95) and jy[i] == _host_file(harness, py[i], 95)


def test_incompressible_planes_grow_the_stream_buffer(dfx, harness):
    """Bounded planes that ARE noise: Farneback without its window (winSize 1, no pyramid) on unrelated noise frames gives a
    flow that is noise pixel by pixel, 5.7 bits per pixel at quality 85 — above the 4 bits per pixel (+ 64 KB) the stream
    buffer is first sized for, below the 8 of the caller's per-file buffers.  The device encoder grows its stream buffers
    from the scan pass's measurement and writes the host encoder's bytes — for the FlowBuffer entry points (two batches:
    the regrown buffers serve both staging parities; blocking and submit form) and for dfx_encode_jpeg."""
    rng = np.random.default_rng(11)
    w, h = 512, 320
    frames = [rng.integers(0, 256, (h, w)).astype(np.uint8) for _ in range(6)]
    kw = dict(max_batch=3, farn_win_size=1, farn_num_levels=0, farn_num_iters=2)
    with dfx.FlowEngine(w, h, "farn", **kw) as eng:
        px, py = eng.calc_optflows_u8(frames, 1, 8)
        jx, jy = eng.calc_optflows_jpeg(frames, 1, 8, 85)
        bpp = 8.0 * sum(len(f) for f in jx + jy) / (len(jx + jy) * w * h)
        assert 4.6 < bpp < 7.5, f"the case is meant to exceed the first buffer size and fit the files' buffers ({bpp:.2f} bpp)"
        for i in range(5):
            assert jx[i] == _host_file(harness, px[i], 85) and jy[i] == _host_file(harness, py[i], 85), i
    with dfx.FlowEngine(w, h, "farn", **kw) as eng:  # a fresh engine: the first FlowBuffer it ever codes overflows
        files = eng.encode_jpeg(px[:3], 85)
        assert all(f == _host_file(harness, p, 85) for f, p in zip(files, px[:3]))


def test_submit_form_and_capacity_error(dfx, harness):
    L = dfx.load_library()
    w, h, n = 320, 240, 5
    frames = [np.ascontiguousarray(f) for f in SynthClip(w, h, 5).frames(n)]
    m = n - 1
    with dfx.FlowEngine(w, h, "farn", max_batch=2) as eng:
        px, py = eng.calc_optflows_u8(frames, 1, 20)
        cap = int(L.dfx_jpeg_capacity(eng._h))
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        outs = []
        for _ in range(2):  # two FlowBuffers in flight
            bx = [np.zeros(cap, np.uint8) for _ in range(m)]
            by = [np.zeros(cap, np.uint8) for _ in range(m)]
            sx, sy, t = (C.c_uint32 * m)(), (C.c_uint32 * m)(), C.c_uint64(0)
            rc = L.dfx_submit_batch_jpeg(eng._h, fp, w, n, 1, -20.0, 20.0, 95, (C.c_void_p * m)(*[b.ctypes.data for b in bx]),
                                         (C.c_void_p * m)(*[b.ctypes.data for b in by]), cap, sx, sy, C.byref(t))
            assert rc == 0, L.dfx_last_error(eng._h)
            outs.append((bx, by, sx, sy, t.value))
        for bx, by, sx, sy, t in outs:
            assert L.dfx_wait(eng._h, t) == 0
            for i in range(m):
                assert bx[i][:sx[i]].tobytes() == _host_file(harness, px[i], 95)
                assert by[i][:sy[i]].tobytes() == _host_file(harness, py[i], 95)
        # a capacity that may not hold a file is an error, not a truncated file — the status that means "encode this
        # FlowBuffer on the host" (4 = DFX_ERR_UNSUPPORTED), reported by the call itself, also in the submit form (a deferred
        # tail could only fail the whole run: ADVICE r3)
        small = 700
        bx = [np.zeros(small, np.uint8) for _ in range(m)]
        by = [np.zeros(small, np.uint8) for _ in range(m)]
        sx, sy = (C.c_uint32 * m)(), (C.c_uint32 * m)()
        rc = L.dfx_calc_batch_jpeg(eng._h, fp, w, n, 1, -20.0, 20.0, 95, (C.c_void_p * m)(*[b.ctypes.data for b in bx]),
                                   (C.c_void_p * m)(*[b.ctypes.data for b in by]), small, sx, sy)
        assert rc == 4 and b"jpg_capacity" in L.dfx_last_error(eng._h)
        t = C.c_uint64(0)
        rc = L.dfx_submit_batch_jpeg(eng._h, fp, w, n, 1, -20.0, 20.0, 95, (C.c_void_p * m)(*[b.ctypes.data for b in bx]),
                                     (C.c_void_p * m)(*[b.ctypes.data for b in by]), small, sx, sy, C.byref(t))
        assert rc == 4 and b"jpg_capacity" in L.dfx_last_error(eng._h)
        assert L.dfx_wait(eng._h, 0) == 0  # nothing pending, nothing failed later
        # a rejected call consumes a pending dfx_next_segments declaration (it applies to the NEXT call only)
        seg = (C.c_int * 2)(3, 2)
        assert L.dfx_next_segments(eng._h, seg, 2) == 0
        assert L.dfx_calc_batch_u8(eng._h, fp, w, n, 1, -20.0, 20.0, None, None, w) == 1  # NULL plane arrays: rejected
        px2, py2 = eng.calc_optflows_u8(frames, 1, 20)  # an unrelated call afterwards pairs as one clip again
        assert len(px2) == m and all(np.array_equal(a, b) for a, b in zip(px2, px))


def test_golden_files(dfx):
    """tests/golden/jpeg_golden.npz — files written by libjpeg-turbo (tests/golden/make_jpeg_golden.py): the device
    encoder on its own (dfx_encode_jpeg) must write those very bytes — smooth, ragged-edge + saturated, busy, noise,
    flow-like planes down to 1x1; q 95 / 50 / 100 / 10."""
    import os

    from tests.golden.make_jpeg_golden import CASES, QUALITIES

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_golden.npz"))
    for name in CASES:
        plane = g[name + "_plane"]
        h, w = plane.shape
        with dfx.FlowEngine(w, h, "farn") as eng:
            for q in QUALITIES:
                got = eng.encode_jpeg([plane, plane[::-1].copy(), plane], q)
                want = g[f"{name}_q{q}_file"].tobytes()
                assert got[0] == want and got[2] == want, (name, q)
                if w * h >= 1000 and q >= 95:
                    assert got[1] != want  # the flipped plane in between is a different file
