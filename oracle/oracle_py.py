"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package (denseflow_amd/) never imports this module.  PARITY UNPINNED: see oracle_common.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_QUANT_PATH = os.path.join(_HERE, "_ref", "libref_quant.so")
REFERENCE_ROOT = "/root/reference"

MAX_SCALES, MAX_WARPS, MAX_CHECKS = 16, 16, 8192


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds)."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if os.path.exists(os.path.join(REFERENCE_ROOT, "src", "common.cpp")):
        # the reference's own convertFlowToImage, compiled where it lies (absent on the GPU box: the prebuilt
        # oracle/_ref/libref_quant.so travels with the snapshot)
        r = subprocess.run(["make", "-C", _HERE, "ref", "REF=" + REFERENCE_ROOT], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle/_ref build failed:\n" + r.stdout + r.stderr)
    return _LIB_PATH


class Tvl1Params(C.Structure):
    _fields_ = [
        ("tau", C.c_double),
        ("lambda_", C.c_double),
        ("theta", C.c_double),
        ("nscales", C.c_int),
        ("warps", C.c_int),
        ("epsilon", C.c_double),
        ("iterations", C.c_int),
        ("scale_step", C.c_double),
        ("gamma", C.c_double),
    ]


class Tvl1Trace(C.Structure):
    _fields_ = [
        ("nscales", C.c_int),
        ("w", C.c_int * MAX_SCALES),
        ("h", C.c_int * MAX_SCALES),
        ("iters", (C.c_int * MAX_WARPS) * MAX_SCALES),
        ("n_checks", C.c_int),
        ("chk_level", C.c_int * MAX_CHECKS),
        ("chk_warp", C.c_int * MAX_CHECKS),
        ("chk_n", C.c_int * MAX_CHECKS),
        ("chk_err", C.c_double * MAX_CHECKS),
    ]

    def iters_table(self):
        return [[self.iters[s][w] for w in range(MAX_WARPS)] for s in range(self.nscales)]

    def checks(self):
        n = min(self.n_checks, MAX_CHECKS)
        return [(self.chk_level[i], self.chk_warp[i], self.chk_n[i], self.chk_err[i]) for i in range(n)]


_lib = None
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    _bind_all(L)
    _lib = L
    return L


_FMA_LIB_PATH = os.path.join(_HERE, "_native", "liboracle_fma.so")


def build_fma() -> str:
    """The same oracle sources with FMA contraction allowed (-ffp-contract=fast -mfma): what nvcc does by default for
    the real cudaoptflow kernels.  A rounding-only variant of the oracle, used to anchor the tolerance bands
    of tests/flow_stats.py in live data and by scripts/oracle_variants.py; never the parity oracle."""
    srcs = [os.path.join(_HERE, f) for f in ("tvl1_oracle.c", "farneback_oracle.c", "brox_oracle.c", "quant_oracle.c",
                                              "prepare_oracle.c", "cpu_tvl1_baseline.c")]
    if not os.path.exists(_FMA_LIB_PATH) or any(os.path.getmtime(x) > os.path.getmtime(_FMA_LIB_PATH) for x in srcs):
        os.makedirs(os.path.dirname(_FMA_LIB_PATH), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=fast", "-mfma",
                        "-D_GNU_SOURCE", "-o", _FMA_LIB_PATH] + srcs + ["-lm"], check=True)
    return _FMA_LIB_PATH


class fma_build:
    """with oracle_py.fma_build(): every *_calc of this module runs the FMA-contracted build of the same sources."""

    def __enter__(self):
        global _lib
        self._old = lib()
        L = C.CDLL(build_fma())
        _bind_all(L)
        _lib = L
        return self

    def __exit__(self, *a):
        global _lib
        _lib = self._old


def _bind_all(L):
    L.orc_set_variant.argtypes = [C.c_int]
    L.orc_get_variant.restype = C.c_int
    L.orc_set_brox_omega.argtypes = [C.c_float]
    L.orc_get_brox_omega.restype = C.c_float
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_get_max_threads.restype = C.c_int
    L.orc_tvl1_default_params.argtypes = [C.POINTER(Tvl1Params)]
    L.orc_tvl1_calc.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Tvl1Params), _f32p,
                                C.POINTER(Tvl1Trace)]
    L.orc_tvl1_calc.restype = C.c_int
    L.orc_resize_linear.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.orc_tvl1_centered_gradient.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p]
    L.orc_tvl1_warp_backward.argtypes = [_f32p] * 6 + [C.c_int, C.c_int] + [_f32p] * 5
    L.orc_tvl1_estimate_u.argtypes = [_f32p] * 10 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    L.orc_tvl1_estimate_u.restype = C.c_double
    L.orc_tvl1_estimate_dual.argtypes = [_f32p] * 6 + [C.c_int, C.c_int, C.c_float]
    L.orc_tvl1_proc_one_scale.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.POINTER(Tvl1Params),
                                          C.c_int, C.POINTER(Tvl1Trace)]
    if hasattr(L, "orc_farneback_calc"):
        L.orc_farneback_default_params.argtypes = [C.POINTER(FarnebackParams)]
        L.orc_farneback_calc.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int,
                                         C.POINTER(FarnebackParams), _f32p]
        L.orc_farneback_calc.restype = C.c_int
    if hasattr(L, "orc_brox_calc"):
        L.orc_brox_default_params.argtypes = [C.POINTER(BroxParams)]
        L.orc_brox_calc.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.POINTER(BroxParams), _f32p]
        L.orc_brox_calc.restype = C.c_int
        L.orc_brox_pyramid_sizes.argtypes = [C.c_int, C.c_int, C.POINTER(BroxParams), C.POINTER(C.c_int), C.c_int]
        L.orc_brox_pyramid_sizes.restype = C.c_int
    if hasattr(L, "orc_prepare_frame"):
        L.orc_prepare_frame.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.orc_resize_u8.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.orc_bgr2gray.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
    if hasattr(L, "orc_flow_to_u8"):
        L.orc_flow_to_u8.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, C.c_double, _u8p, _u8p]
    if hasattr(L, "cpu_tvl1_calc"):
        _bind_cpu_tvl1(L)
    return L


# reading variants (oracle_common.h): flags for `variant(...)`
VAR_TVL1_BREAK_BEFORE_DUAL, VAR_TVL1_SUM_FLOAT, VAR_TVL1_SQRT_HYPOT = 1, 2, 4
VAR_FARN_SIGMA0_COMPUTED, VAR_BROX_JACOBI, VAR_BROX_CONVERT_DOUBLE = 8, 16, 32
VAR_TVL1_LIBM_HYPOT = 64


class variant:
    """with oracle_py.variant(flags, brox_omega=...): the oracle evaluates an alternative reading of upstream."""

    def __init__(self, flags: int = 0, brox_omega: float = 0.0):
        self.flags, self.omega = int(flags), float(brox_omega)

    def __enter__(self):
        L = lib()
        self._old = (L.orc_get_variant(), L.orc_get_brox_omega())
        L.orc_set_variant(self.flags)
        L.orc_set_brox_omega(self.omega)
        return self

    def __exit__(self, *a):
        L = lib()
        L.orc_set_variant(self._old[0])
        L.orc_set_brox_omega(self._old[1])


def _pick_threads(h: int, w: int, threads):
    """One thread per ~16k pixels unless the caller fixes it (bench: all cores).  The row loops are
    short; on a 256-thread host the fork/join cost of every tiny parallel region dominates otherwise."""
    L = lib()
    if threads is None:
        threads = max(1, min((h * w) // 16384, os.cpu_count() or 1, 128))  # never more than one socket's worth
    L.orc_set_num_threads(int(threads))
    return L.orc_get_max_threads()


def tvl1_default_params() -> Tvl1Params:
    p = Tvl1Params()
    lib().orc_tvl1_default_params(C.byref(p))
    return p


def tvl1_calc(frame0: np.ndarray, frame1: np.ndarray, params: Tvl1Params | None = None, want_trace: bool = False,
              threads=None):
    """cv::cuda::OpticalFlowDual_TVL1::calc restatement. Returns flow (H,W,2) [, trace]."""
    f0 = np.ascontiguousarray(frame0, dtype=np.uint8)
    f1 = np.ascontiguousarray(frame1, dtype=np.uint8)
    assert f0.shape == f1.shape and f0.ndim == 2
    h, w = f0.shape
    _pick_threads(h, w, threads)
    flow = np.empty((h, w, 2), dtype=np.float32)
    trace = Tvl1Trace()
    rc = lib().orc_tvl1_calc(f0, w, f1, w, w, h, C.byref(params) if params is not None else None, flow,
                             C.byref(trace))
    if rc != 0:
        raise ValueError("orc_tvl1_calc rejected the parameters")
    return (flow, trace) if want_trace else flow


def resize_linear(src: np.ndarray, dw: int, dh: int, ifx: float, ify: float) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.float32)
    sh, sw = src.shape
    dst = np.empty((dh, dw), dtype=np.float32)
    lib().orc_resize_linear(src, sw, sh, dst, dw, dh, ifx, ify)
    return dst


# ------------------------------------------------------------------------------ Farneback

class FarnebackParams(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int),
        ("pyr_scale", C.c_double),
        ("fast_pyramids", C.c_int),
        ("win_size", C.c_int),
        ("num_iters", C.c_int),
        ("poly_n", C.c_int),
        ("poly_sigma", C.c_double),
        ("flags", C.c_int),
    ]


def farneback_default_params() -> FarnebackParams:
    p = FarnebackParams()
    lib().orc_farneback_default_params(C.byref(p))
    return p


def farneback_calc(frame0: np.ndarray, frame1: np.ndarray, params: FarnebackParams | None = None,
                   threads=None) -> np.ndarray:
    f0 = np.ascontiguousarray(frame0, dtype=np.uint8)
    f1 = np.ascontiguousarray(frame1, dtype=np.uint8)
    assert f0.shape == f1.shape and f0.ndim == 2
    h, w = f0.shape
    _pick_threads(h, w, threads)
    flow = np.empty((h, w, 2), dtype=np.float32)
    rc = lib().orc_farneback_calc(f0, w, f1, w, w, h, C.byref(params) if params is not None else None, flow)
    if rc != 0:
        raise ValueError("orc_farneback_calc rejected the parameters")
    return flow


# ------------------------------------------------------------------------------ Brox

class BroxParams(C.Structure):
    _fields_ = [
        ("alpha", C.c_float),
        ("gamma", C.c_float),
        ("scale_factor", C.c_float),
        ("inner_iterations", C.c_int),
        ("outer_iterations", C.c_int),
        ("solver_iterations", C.c_int),
    ]


def brox_default_params() -> BroxParams:
    p = BroxParams()
    lib().orc_brox_default_params(C.byref(p))
    return p


def brox_pyramid_sizes(w: int, h: int, params: BroxParams | None = None):
    p = params if params is not None else brox_default_params()
    buf = (C.c_int * 256)()
    n = lib().orc_brox_pyramid_sizes(w, h, C.byref(p), buf, 128)
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]


def brox_calc(frame0: np.ndarray, frame1: np.ndarray, params: BroxParams | None = None, threads=None) -> np.ndarray:
    """cv::cuda::BroxOpticalFlow restatement as DEFINED in oracle/brox_oracle.h (u8 frames in, the
    1/255 pre-scale of src/denseflow_gpu.cpp:332-333 is part of it)."""
    f0 = np.ascontiguousarray(frame0, dtype=np.uint8)
    f1 = np.ascontiguousarray(frame1, dtype=np.uint8)
    assert f0.shape == f1.shape and f0.ndim == 2
    h, w = f0.shape
    _pick_threads(h, w, threads)
    flow = np.empty((h, w, 2), dtype=np.float32)
    rc = lib().orc_brox_calc(f0, w, f1, w, w, h, C.byref(params) if params is not None else None, flow)
    if rc != 0:
        raise ValueError("orc_brox_calc rejected the parameters")
    return flow


# ---------------------------------------------------------------------------------------------- flow bounding
def flow_to_u8(flow: np.ndarray, lower: float, upper: float):
    """convertFlowToImage (reference src/common.cpp:4-16) on one (H, W, 2) float32 flow -> (img_x, img_y)."""
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    h, w, _ = flow.shape
    img_x = np.empty((h, w), np.uint8)
    img_y = np.empty((h, w), np.uint8)
    lib().orc_flow_to_u8(flow, w, h, float(lower), float(upper), img_x, img_y)
    return img_x, img_y


_REF_PNG_PATH = os.path.join(_HERE, "_ref", "libref_png.so")
_ref_png = None


def flow_to_png_planes(flow: np.ndarray):
    """The -st=png scheme (quant_oracle.h: orc_flow_to_png_planes).  Returns (plane_x, plane_y, (bound_x, bound_y), bgr)."""
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    x, y = np.empty((h, w), np.uint8), np.empty((h, w), np.uint8)
    bgr = np.empty((h, w, 3), np.uint8)
    b = (C.c_double * 2)()
    fn = lib().orc_flow_to_png_planes
    fn.argtypes = [_f32p, C.c_int, C.c_int, _u8p, _u8p, C.POINTER(C.c_double), _u8p]
    fn.restype = None
    fn(flow, w, h, x, y, b, bgr)
    return x, y, (b[0], b[1]), bgr


def ref_png_available() -> bool:
    return os.path.exists(_REF_PNG_PATH)


def ref_flow_to_png_image(flow: np.ndarray) -> np.ndarray:
    """The same through the reference's own source lines (oracle/_ref/libref_png.so, `make -C oracle ref`): (h, w, 3)."""
    global _ref_png
    if _ref_png is None:
        R = C.CDLL(_REF_PNG_PATH)
        R.ref_convert_flow_to_png_image.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _u8p]
        R.ref_convert_flow_to_png_image.restype = None
        _ref_png = R
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    fx, fy = np.ascontiguousarray(flow[..., 0]), np.ascontiguousarray(flow[..., 1])
    out = np.empty((h, w, 3), np.uint8)
    _ref_png.ref_convert_flow_to_png_image(fx, fy, w, h, out)
    return out


_ref_quant = None


def ref_quant_available() -> bool:
    return os.path.exists(_REF_QUANT_PATH)


def ref_flow_to_u8(flow: np.ndarray, lower: float, upper: float):
    """The same through the reference's own source lines (oracle/_ref/libref_quant.so, `make -C oracle ref`)."""
    global _ref_quant
    if _ref_quant is None:
        R = C.CDLL(_REF_QUANT_PATH)
        R.ref_convert_flow_to_image.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_double, C.c_double, _u8p, _u8p]
        _ref_quant = R
    flow = np.asarray(flow, dtype=np.float32)
    h, w, _ = flow.shape
    fx = np.ascontiguousarray(flow[..., 0])
    fy = np.ascontiguousarray(flow[..., 1])
    img_x = np.empty((h, w), np.uint8)
    img_y = np.empty((h, w), np.uint8)
    _ref_quant.ref_convert_flow_to_image(fx, fy, w, h, float(lower), float(upper), img_x, img_y)
    return img_x, img_y


# ---------------------------------------------------------------------------------------------- frame preparation
def prepare_frame(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """The loader's cvtColor(BGR2GRAY) + cv::resize (reference src/denseflow_gpu.cpp:163, :169) for one frame:
    src (H, W) gray or (H, W, 3) BGR uint8 -> (dh, dw) uint8."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    ch = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_prepare_frame(src.reshape(-1), sw, sh, ch, dst, dw, dh)
    return dst


# ---------------------------------------------------------------------------------------------- CPU DualTVL1 baseline
class CpuTvl1Params(C.Structure):
    _fields_ = [
        ("tau", C.c_double),
        ("lambda_", C.c_double),
        ("theta", C.c_double),
        ("nscales", C.c_int),
        ("warps", C.c_int),
        ("epsilon", C.c_double),
        ("inner_iterations", C.c_int),
        ("outer_iterations", C.c_int),
        ("scale_step", C.c_double),
        ("median_filtering", C.c_int),
    ]


class CpuTvl1Stats(C.Structure):
    _fields_ = [
        ("nscales", C.c_int),
        ("w", C.c_int * 16),
        ("h", C.c_int * 16),
        ("inner_iterations", C.c_longlong),
        ("outer_iterations", C.c_longlong),
        ("px_iterations", C.c_double),
    ]


def _bind_cpu_tvl1(L):
    L.cpu_tvl1_default_params.argtypes = [C.POINTER(CpuTvl1Params)]
    L.cpu_tvl1_calc.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.POINTER(CpuTvl1Params), _f32p,
                                C.POINTER(CpuTvl1Stats)]
    L.cpu_tvl1_calc.restype = C.c_int
    L.cpu_tvl1_resize_linear.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_double, C.c_double]
    L.cpu_tvl1_median_blur.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int]
    L.cpu_tvl1_remap_cubic.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p]
    L.cpu_tvl1_cubic_coeffs.argtypes = [C.c_float, C.POINTER(C.c_float * 4)]


_native_cpu_tvl1 = None


def cpu_tvl1_native_lib():
    """oracle/cpu_tvl1_baseline.c compiled ON THIS MACHINE with -O3 -march=native (the CPU
    comparator is built for the host it is timed on).  Falls back to the portable copy inside liboracle.so when
    no compiler is available."""
    global _native_cpu_tvl1
    if _native_cpu_tvl1 is not None:
        return _native_cpu_tvl1
    import hashlib
    import platform

    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln for ln in f if ln.startswith("model name")), platform.processor())
    except OSError:
        model = platform.processor()
    tag = hashlib.sha1(model.encode()).hexdigest()[:10]
    out = os.path.join(_HERE, "_native", f"libcpu_tvl1_{tag}.so")
    src = os.path.join(_HERE, "cpu_tvl1_baseline.c")
    L = None
    try:
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            subprocess.run(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-shared", "-fopenmp",
                            "-ffp-contract=off", "-D_GNU_SOURCE", "-o", out, src, "-lm"], check=True,
                           capture_output=True)
        L = C.CDLL(out)
        _bind_cpu_tvl1(L)
    except Exception:
        L = lib()
    _native_cpu_tvl1 = L
    return L


def cpu_tvl1_default_params() -> CpuTvl1Params:
    p = CpuTvl1Params()
    lib().cpu_tvl1_default_params(C.byref(p))
    return p


def cpu_tvl1_calc(frame0: np.ndarray, frame1: np.ndarray, params: CpuTvl1Params | None = None, want_stats=False,
                  native: bool = False):
    """CPU cv::optflow::DualTVL1OpticalFlow restatement (oracle/cpu_tvl1_baseline.c) — the timing comparator."""
    f0 = np.ascontiguousarray(frame0, dtype=np.uint8)
    f1 = np.ascontiguousarray(frame1, dtype=np.uint8)
    assert f0.shape == f1.shape and f0.ndim == 2
    h, w = f0.shape
    flow = np.empty((h, w, 2), dtype=np.float32)
    st = CpuTvl1Stats()
    L = cpu_tvl1_native_lib() if native else lib()
    rc = L.cpu_tvl1_calc(f0, w, f1, w, w, h, C.byref(params) if params is not None else None, flow, C.byref(st))
    if rc != 0:
        raise ValueError("cpu_tvl1_calc rejected the parameters")
    return (flow, st) if want_stats else flow
