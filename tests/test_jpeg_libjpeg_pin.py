"""PIN of the JPEG stage (the reference's imencode(".jpg") in encodeFlowMap, /root/reference/src/
common.cpp:56-57) against the real libjpeg.  cv::imencode drives libjpeg(-turbo) with its defaults (baseline, Annex K
tables, JDCT_ISLOW); libjpeg is not in /root/reference, but Pillow ships a libjpeg-turbo and drives it the same way, so
where Pillow is importable the host encoder is compared with it LIVE, byte for byte, over sizes, qualities and contents
(flow-like, saturated, noise, ragged edges, 1x1).  tests/golden/jpeg_golden.npz holds libjpeg-turbo's files for the
boxes without Pillow (tests/test_jpeg_host.py; the device encoder: tests/test_jpeg_gpu.py)."""
import ctypes as C
import io

import numpy as np
import pytest

from tests.test_host_shell import built, harness  # noqa: F401  (fixtures)


def _libjpeg(plane, quality):
    Image = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(plane), "L").save(b, "JPEG", quality=quality)
    return b.getvalue()


def _host(harness, plane, quality, portable=0):
    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    harness.hh_jpeg_force_portable(portable)
    plane = np.ascontiguousarray(plane)
    h, w = plane.shape
    buf = np.zeros(w * h * 3 + 4096, np.uint8)
    n = harness.hh_encode_jpeg(plane.ctypes.data, w, h, quality, buf.ctypes.data, buf.size)
    harness.hh_jpeg_force_portable(0)
    assert n > 0
    return buf[:n].tobytes()


def _contents(rng, w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.clip(128 + 90 * np.sin(xx / 11.0) * np.cos(yy / 5.0) + rng.normal(0, 8, (h, w)), 0, 255).astype(np.uint8)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    flat = np.clip(128 + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)
    flat[: h // 3, : w // 3] = 255  # saturated corners: 0xFF runs in the segment (stuffing), long zero runs
    flat[h // 2:, w // 2:] = 0
    return {"smooth": smooth, "noise": noise, "saturated": flat}


@pytest.mark.parametrize("w,h", [(64, 48), (70, 45), (8, 8), (257, 131), (5, 3), (640, 360), (33, 17), (1, 1), (7, 9), (1920, 1080)])
def test_host_encoder_writes_libjpegs_bytes(harness, w, h):
    rng = np.random.default_rng(w * 7 + h)
    for q in ((95, 50) if w * h > 500000 else (1, 10, 50, 75, 90, 95, 100)):
        for kind, plane in _contents(rng, w, h).items():
            want = _libjpeg(plane, q)
            assert _host(harness, plane, q) == want, (w, h, q, kind, "vector form")
            if w * h <= 100000:
                assert _host(harness, plane, q, portable=1) == want, (w, h, q, kind, "scalar form")


def test_bounded_flow_planes_are_libjpegs_bytes(harness, oracle):
    """The planes the save stage really encodes: Farneback flows of the synthetic clip, bounded at 20 (encodeFlowMap's
    convertFlowToImage, pinned separately in tests/test_quant_*.py)."""
    from denseflow_amd.synth import SynthClip

    w, h = 320, 240
    frames = SynthClip(w, h, 31).frames(3)
    for i in range(2):
        flow = oracle.farneback_calc(frames[i], frames[i + 1])
        for plane in oracle.flow_to_u8(flow, -20.0, 20.0):
            assert _host(harness, plane, 95) == _libjpeg(plane, 95)


def test_reciprocal_quantisation_is_libjpegs_integer_division(harness):
    """dfx_jpeg_quantise (include/dfx_jpeg_tables.h: quotient as the high half of a 32 x 32 product) against libjpeg's
    rule with a plain division, exhaustively: every divisor 8 q, q = 1..255, every transform output in +-40000 (8-bit
    samples give |output| <= 8 * 1024 * 1.39 < 12000)."""
    harness.hh_jpeg_quantise_mismatches.restype = C.c_longlong
    assert harness.hh_jpeg_quantise_mismatches(40000) == 0


def test_random_planes_against_libjpeg(harness):
    """Property form of the live pin: random sizes (1 ... 150 per side: ragged right / bottom blocks of every width), every
    quality 1 ... 100, contents from constant to noise; scalar and vector transforms."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    pytest.importorskip("PIL.Image")

    @settings(max_examples=150, deadline=None)
    @given(w=st.integers(1, 150), h=st.integers(1, 150), q=st.integers(1, 100), kind=st.integers(0, 3), seed=st.integers(0, 2 ** 31))
    def check(w, h, q, kind, seed):
        rng = np.random.default_rng(seed)
        if kind == 0:
            plane = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:
            plane = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        elif kind == 2:
            yy, xx = np.mgrid[0:h, 0:w]
            plane = np.clip(128 + 100 * np.sin(xx / 7.0 + seed % 7) * np.cos(yy / 5.0) + rng.normal(0, 3, (h, w)), 0, 255).astype(np.uint8)
        else:  # saturated: 0 / 255 only (the extreme coefficients)
            plane = (rng.random((h, w)) < 0.5).astype(np.uint8) * 255
        want = _libjpeg(plane, q)
        assert _host(harness, plane, q) == want
        assert _host(harness, plane, q, portable=1) == want

    check()
