"""The host half of the device JPEG encoder (denseflow_amd/csrc/jpeg_host.cpp), without a GPU: file header and byte
stuffing must reproduce the shell's host encoder (src/image_io.cpp, imencodeJpeg) exactly.  A file written by the host
encoder is split into header / entropy-coded segment, the segment is un-stuffed into the plain bit string the device
produces, and dfxi_jpeg_assemble must rebuild the very same file from it."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.test_host_shell import built, harness  # noqa: F401  (fixtures)


def _unstuff(seg: bytes):
    out = bytearray()
    i = 0
    while i < len(seg):
        out.append(seg[i])
        if seg[i] == 0xFF:
            assert seg[i + 1] == 0x00
            i += 1
        i += 1
    return bytes(out)


@pytest.mark.parametrize("w,h,quality", [(64, 48, 95), (70, 45, 95), (8, 8, 50), (257, 131, 100), (5, 3, 10), (640, 360, 95)])
def test_assembly_reproduces_the_host_encoders_file(harness, w, h, quality):
    import denseflow_amd

    L = denseflow_amd.load_library()
    L.dfxi_jpeg_assemble.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_size_t]
    L.dfxi_jpeg_assemble.restype = C.c_size_t
    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(w + h + quality)
    yy, xx = np.mgrid[0:h, 0:w]
    # smooth + noise, and a saturated corner: runs of 0xFF bytes in the segment exercise the stuffing
    gray = np.clip(128 + 90 * np.sin(xx / 11.0) * np.cos(yy / 5.0) + rng.normal(0, 25, (h, w)), 0, 255).astype(np.uint8)
    gray[: h // 3, : w // 3] = 255
    buf = np.zeros(4 << 20, np.uint8)
    n = harness.hh_encode_jpeg(gray.ctypes.data, w, h, quality, buf.ctypes.data, buf.size)
    assert n > 0
    ref = buf[:n].tobytes()
    sos = ref.index(b"\xff\xda")
    header_len = sos + 2 + 8
    assert ref.endswith(b"\xff\xd9")
    plain = _unstuff(ref[header_len:-2])
    # the host encoder pads the last byte with ones; the device reports the exact bit count.  Try every possible
    # count that ends in this byte: the right one reproduces the file (the padding bits are ones either way)
    ok = False
    for pad in range(8):
        bits = len(plain) * 8 - pad
        seg = bytearray(plain)
        if pad:
            if seg[-1] & ((1 << pad) - 1) != (1 << pad) - 1:
                continue
            seg[-1] &= 0xFF ^ ((1 << pad) - 1)  # the device leaves the unused bits zero
        seg = bytes(seg)
        out = np.zeros(n + 64, np.uint8)
        m = L.dfxi_jpeg_assemble(w, h, quality, seg, bits, out.ctypes.data, out.size)
        if m == n and out[:m].tobytes() == ref:
            ok = True
            break
    assert ok, "no bit count reproduces the host encoder's file"
    # capacity check: one byte too few is refused
    out = np.zeros(n - 1, np.uint8)
    assert L.dfxi_jpeg_assemble(w, h, quality, seg, bits, out.ctypes.data, out.size) == 0


def test_simd_and_portable_transforms_are_now_bit_identical(harness):
    """Both host forms of the transform (eight lines at a time on vector types, and scalar) are libjpeg's integer
    JDCT_ISLOW: the files must be IDENTICAL, not just close."""
    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(4)
    for (w, h) in [(64, 64), (70, 45), (257, 131), (9, 17)]:
        gray = rng.integers(0, 256, (h, w)).astype(np.uint8)
        files = []
        for portable in (0, 1):
            harness.hh_jpeg_force_portable(portable)
            buf = np.zeros(1 << 20, np.uint8)
            n = harness.hh_encode_jpeg(gray.ctypes.data, w, h, 95, buf.ctypes.data, buf.size)
            files.append(buf[:n].tobytes())
        harness.hh_jpeg_force_portable(0)
        assert files[0] == files[1], (w, h)


def test_host_encoder_reproduces_the_golden_files(harness):
    """The shell's encoder against tests/golden/jpeg_golden.npz — files written by libjpeg-turbo itself (Pillow;
    tests/golden/make_jpeg_golden.py), the library behind the reference's imencode(".jpg").  The GPU suite holds the
    device encoder to the same bytes."""
    import os

    from tests.golden.make_jpeg_golden import CASES, QUALITIES

    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_golden.npz"))
    assert bytes(g["libjpeg"]).startswith(b"libjpeg-turbo")
    for name in CASES:
        plane = np.ascontiguousarray(g[name + "_plane"])
        for q in QUALITIES:
            for portable in (0, 1):
                harness.hh_jpeg_force_portable(portable)
                buf = np.zeros(1 << 20, np.uint8)
                n = harness.hh_encode_jpeg(plane.ctypes.data, plane.shape[1], plane.shape[0], q, buf.ctypes.data, buf.size)
                assert buf[:n].tobytes() == g[f"{name}_q{q}_file"].tobytes(), (name, q, portable)
    harness.hh_jpeg_force_portable(0)


def test_assembly_fuzz_under_sanitizers(tmp_path):
    """jpeg_assemble (denseflow_amd/csrc/jpeg_host.cpp) on 12 000 random bit strings (all-0xFF, noise, mostly-0xFF), compiled
    with AddressSanitizer + UBSan: destination buffers of exactly the file size (the capacity check is exact), one byte less
    refused, correct stuffing and EOI (tests/jpeg_assemble_fuzz.cpp)."""
    import shutil
    import subprocess

    from tests.test_host_shell import ROOT

    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / "jpeg_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                        os.path.join(ROOT, "tests", "jpeg_assemble_fuzz.cpp"),
                        os.path.join(ROOT, "denseflow_amd", "csrc", "jpeg_host.cpp")], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr.lower():
        pytest.skip("sanitizer build not available: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "ok 12000" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
