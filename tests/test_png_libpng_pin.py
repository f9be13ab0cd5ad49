"""PIN of the PNG writer (the reference's imencode(".png") at the end of encodeFlowMapPng, /root/reference/src/common.cpp:
66-71) against the real libpng: the system's libpng16 driven through ctypes with the calls OpenCV's PngEncoder makes
(tests/libpng_ref.py), byte for byte, live where the library loads and against tests/golden/png_golden.npz (files that
libpng wrote) everywhere.  Lossless container, so pixel parity was never in doubt; this pins the BYTES (filter choice,
zlib level / strategy / window, IDAT chunking)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.test_host_shell import built, harness  # noqa: F401  (fixtures)


def _mine(harness, img):
    harness.hh_encode_png.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else 3
    buf = np.zeros(w * h * ch * 2 + 4096, np.uint8)
    n = harness.hh_encode_png(img.ctypes.data, w, h, ch, buf.ctypes.data, buf.size)
    assert n > 0
    return buf[:n].tobytes()


def test_golden_files_written_by_libpng(harness):
    from tests.golden.make_png_golden import CASES

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "png_golden.npz"))
    assert bytes(g["libpng"]).startswith(b"libpng 1.6")
    for name in CASES:
        assert _mine(harness, g[name + "_image"]) == g[name + "_file"].tobytes(), name


@pytest.mark.parametrize("w,h", [(1, 1), (1, 9), (9, 1), (2, 2), (5, 3), (33, 17), (73, 74), (74, 74), (127, 129), (181, 90),
                                 (182, 90), (257, 131), (640, 360), (1920, 1080)])
def test_live_against_libpng(harness, w, h):
    """Sizes straddle libpng's small-image rules: <= 16384 filtered bytes (window shrinking), one-pixel-wide images (no
    SUB), compressed streams longer than one 8192-byte IDAT."""
    from tests import libpng_ref

    if libpng_ref.load() is None:
        pytest.skip("no libpng16 on this machine (the golden test covers it)")
    rng = np.random.default_rng(w * 3 + h)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.clip(128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 0, 255)
    for ch in (1, 3):
        shape = (h, w) if ch == 1 else (h, w, 3)
        for img in (rng.integers(0, 256, shape, dtype=np.uint8),
                    (base if ch == 1 else np.stack([base, 255 - base, base / 2], -1)).astype(np.uint8),
                    np.full(shape, 128, np.uint8)):
            assert _mine(harness, img) == libpng_ref.imencode_png(img), (w, h, ch)


def test_encoded_flow_png_is_libpngs_file(harness, oracle):
    """encodeFlowMapPng end to end on the host shell: the adaptive-bound BGR image of a Farneback flow
    (convertFlowToPngImage, src/common.cpp:18-46) through imencodePng == libpng's file for the image it decodes to."""
    import io

    from PIL import Image

    from denseflow_amd.synth import SynthClip
    from tests import libpng_ref

    if libpng_ref.load() is None:
        pytest.skip("no libpng16 on this machine")
    w, h = 160, 120
    frames = SynthClip(w, h, 9).frames(2)
    flow = oracle.farneback_calc(frames[0], frames[1])
    fx, fy = np.ascontiguousarray(flow[..., 0]), np.ascontiguousarray(flow[..., 1])
    buf = np.zeros(w * h * 8, np.uint8)
    harness.hh_encode_flow_png.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = harness.hh_encode_flow_png(fx.ctypes.data, fy.ctypes.data, w, h, buf.ctypes.data, buf.size)
    assert n > 0
    rgb = np.array(Image.open(io.BytesIO(buf[:n].tobytes())))
    assert rgb.shape == (h, w, 3)
    assert buf[:n].tobytes() == libpng_ref.imencode_png(rgb[..., ::-1])


def test_random_images_against_libpng(harness):
    """Property form of the live pin: random sizes (1 ... 200 per side, so both libpng's small-image rules and multi-IDAT
    streams occur), gray and BGR, contents from constant to noise."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from tests import libpng_ref

    if libpng_ref.load() is None:
        pytest.skip("no libpng16 on this machine")

    @settings(max_examples=120, deadline=None)
    @given(w=st.integers(1, 200), h=st.integers(1, 200), ch=st.sampled_from([1, 3]), kind=st.integers(0, 3),
           seed=st.integers(0, 2 ** 31))
    def check(w, h, ch, kind, seed):
        rng = np.random.default_rng(seed)
        shape = (h, w) if ch == 1 else (h, w, 3)
        if kind == 0:
            img = rng.integers(0, 256, shape, dtype=np.uint8)
        elif kind == 1:
            img = np.full(shape, int(rng.integers(0, 256)), np.uint8)
        elif kind == 2:  # smooth ramps with a little noise: long and short RLE runs mixed
            yy, xx = np.mgrid[0:h, 0:w]
            base = (xx * 3 + yy * 2) % 256 + rng.integers(0, 2, (h, w))
            img = np.clip(base if ch == 1 else np.stack([base, 255 - base, base // 2], -1), 0, 255).astype(np.uint8)
        else:  # mostly constant with sparse spikes
            img = np.full(shape, 128, np.uint8)
            mask = rng.random(shape[:2]) < 0.02
            img[mask] = 255
        assert _mine(harness, img) == libpng_ref.imencode_png(img)

    check()
