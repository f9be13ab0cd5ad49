"""ctypes binding of include/dfx.h.

Mirrors the reference operator interface for the hot path: `FlowEngine.calc_optflows` takes the
gray frames of one FlowBuffer and a step and returns the list of CV_32FC2-shaped flows exactly as
DenseFlow::calc_optflows_imp does (/root/reference is not needed at run time; the behaviour is the
one at src/denseflow_gpu.cpp:307-342).  Errors carry the reference's message texts.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB_DIR = os.path.join(_HERE, "lib")
_LIB = os.path.join(_LIB_DIR, "libdfx.so")
_CSRC = os.path.join(_HERE, "csrc")

DFX_MAX_LEVELS = 32
DFX_MAX_WARPS = 16

ALGO_TVL1, ALGO_FARN, ALGO_BROX = 0, 1, 2
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_NV_DISABLED, ERR_UNKNOWN_ALGO = range(7)


class DfxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class DfxParams(C.Structure):
    _fields_ = [
        ("tvl1_tau", C.c_double),
        ("tvl1_lambda", C.c_double),
        ("tvl1_theta", C.c_double),
        ("tvl1_nscales", C.c_int),
        ("tvl1_warps", C.c_int),
        ("tvl1_epsilon", C.c_double),
        ("tvl1_iterations", C.c_int),
        ("tvl1_scale_step", C.c_double),
        ("farn_num_levels", C.c_int),
        ("farn_pyr_scale", C.c_double),
        ("farn_win_size", C.c_int),
        ("farn_num_iters", C.c_int),
        ("farn_poly_n", C.c_int),
        ("farn_poly_sigma", C.c_double),
        ("farn_flags", C.c_int),
        ("brox_alpha", C.c_float),
        ("brox_gamma", C.c_float),
        ("brox_scale_factor", C.c_float),
        ("brox_inner_iterations", C.c_int),
        ("brox_outer_iterations", C.c_int),
        ("brox_solver_iterations", C.c_int),
        ("max_batch", C.c_int),
        ("impl", C.c_int),
        ("tvl1_fuse_k", C.c_int),
        ("tvl1_math", C.c_int),
        ("variant", C.c_int),
        ("step_group", C.c_int),
        ("blocking_sync", C.c_int),
    ]


# dfx_params.variant bits (include/dfx.h): cross-check / measurement forms of the tuned kernels, all bit-identical
VAR_TVL1_CLASSIC_GEOM, VAR_TVL1_WARP_IN_STEP = 0x01, 0x02
VAR_FARN_EVAL_ZERO_TAPS, VAR_FARN_POLY_ONE_ROW, VAR_FARN_M_IN_HBM = 0x04, 0x08, 0x10
VAR_TVL1_WARP_GATHER = 0x20
VAR_TVL1_NO_HEAD = 0x40
VAR_BROX_SOR_PROGRESS = 0x80
VAR_BROX_SOR_PER_TILE = 0x100


class DfxStats(C.Structure):
    _fields_ = [
        ("pairs", C.c_uint64),
        ("batch", C.c_int),
        ("kernel_launches", C.c_uint64),
        ("noop_steps", C.c_uint64),
        ("device_ms", C.c_double),
        ("step_ms", C.c_double),
        ("step_launches", C.c_uint64),
        ("level_ms", C.c_double * DFX_MAX_LEVELS),
        ("level_launches", C.c_uint64 * DFX_MAX_LEVELS),
        ("algorithmic_bytes", C.c_double),
        ("step_algorithmic_bytes", C.c_double),
        ("levels", C.c_int),
        ("level_w", C.c_int * DFX_MAX_LEVELS),
        ("level_h", C.c_int * DFX_MAX_LEVELS),
        ("tvl1_iters", (C.c_int * DFX_MAX_WARPS) * DFX_MAX_LEVELS),
        ("tvl1_checks", C.c_int),
        ("tvl1_total_iters", C.c_uint64),
        ("tvl1_px_iters", C.c_double),
        ("tvl1_lane_iters", C.c_double),
    ]

    def iters_table(self):
        return [[self.tvl1_iters[s][w] for w in range(DFX_MAX_WARPS)] for s in range(self.levels)]


# ------------------------------------------------------------------------------------------ build

def library_path() -> str:
    return _LIB


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into denseflow_amd/lib/libdfx.so (cross-compiles without a GPU)."""
    srcs = sorted(
        os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".hip") or f.endswith(".cpp")
    )
    inc = os.path.join(_ROOT, "include")
    deps = srcs + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")] + [
        os.path.join(inc, f) for f in os.listdir(inc) if f.startswith("dfx") and f.endswith(".h")
    ]
    if not force and os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(d) for d in deps):
        return _LIB
    os.makedirs(_LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [
        hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
        "-I" + os.path.join(_ROOT, "include"), "-o", _LIB,
    ] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return _LIB


_lib = None


def load_library():
    """dlopen libdfx.so.  torch (if present) is imported first so both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise DfxError(ERR_NO_DEVICE, f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # torch bundles its own libamdhip64.so (HIP 7.0 in torch 2.10+rocm7.0); importing it first makes the loader reuse that
    # copy, which a process that also holds torch tensors needs (one runtime per process).  A torch-free process
    # (DFX_NO_TORCH=1: binding-level switch, the library itself reads no environment) gets the system runtime libdfx.so
    # was linked against (/opt/rocm, HIP 7.2) — what a C++ caller such as build/denseflow runs on.  The two runtimes differ
    # in how they execute device-to-host copies (shader blit vs SDMA: DESIGN.md section 5).
    if os.environ.get("DFX_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the library itself
            pass
    L = C.CDLL(os.environ.get("DFX_LIBRARY", _LIB))  # DFX_LIBRARY: A/B a differently built libdfx.so
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    L.dfx_device_count.restype = i
    L.dfx_default_params.argtypes = [C.POINTER(DfxParams)]
    L.dfx_algo_from_name.argtypes = [C.c_char_p, C.POINTER(i)]
    L.dfx_algo_from_name.restype = i
    L.dfx_algo_error_message.argtypes = [i, C.c_char_p, C.c_char_p, sz]
    L.dfx_algo_error_message.restype = C.c_char_p
    L.dfx_create.argtypes = [C.POINTER(vp), i, i, i, i, C.POINTER(DfxParams)]
    L.dfx_create.restype = i
    L.dfx_calc.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    L.dfx_calc.restype = i
    L.dfx_calc_batch.argtypes = [vp, C.POINTER(vp), sz, i, i, C.POINTER(vp), sz]
    L.dfx_calc_batch.restype = i
    L.dfx_calc_batch_device.argtypes = [vp, vp, sz, sz, i, i, vp, sz]
    L.dfx_calc_batch_device.restype = i
    L.dfx_calc_batch_u8.argtypes = [vp, C.POINTER(vp), sz, i, i, C.c_double, C.c_double, C.POINTER(vp), C.POINTER(vp), sz]
    L.dfx_calc_batch_u8.restype = i
    L.dfx_submit_batch.argtypes = [vp, C.POINTER(vp), sz, i, i, C.POINTER(vp), sz, C.POINTER(C.c_uint64)]
    L.dfx_submit_batch.restype = i
    L.dfx_submit_batch_u8.argtypes = [vp, C.POINTER(vp), sz, i, i, C.c_double, C.c_double, C.POINTER(vp), C.POINTER(vp),
                                      sz, C.POINTER(C.c_uint64)]
    L.dfx_submit_batch_u8.restype = i
    u32p = C.POINTER(C.c_uint32)
    L.dfx_calc_batch_jpeg.argtypes = [vp, C.POINTER(vp), sz, i, i, C.c_double, C.c_double, i, C.POINTER(vp),
                                      C.POINTER(vp), sz, u32p, u32p]
    L.dfx_calc_batch_jpeg.restype = i
    L.dfx_submit_batch_jpeg.argtypes = [vp, C.POINTER(vp), sz, i, i, C.c_double, C.c_double, i, C.POINTER(vp),
                                        C.POINTER(vp), sz, u32p, u32p, C.POINTER(C.c_uint64)]
    L.dfx_submit_batch_jpeg.restype = i
    L.dfx_encode_jpeg.argtypes = [vp, C.POINTER(vp), sz, i, i, C.POINTER(vp), sz, u32p]
    L.dfx_encode_jpeg.restype = i
    L.dfx_jpeg_capacity.argtypes = [vp]
    L.dfx_jpeg_capacity.restype = sz
    L.dfx_next_segments.argtypes = [vp, C.POINTER(C.c_int), i]
    L.dfx_next_segments.restype = i
    L.dfx_wait.argtypes = [vp, C.c_uint64]
    L.dfx_wait.restype = i
    L.dfx_calc_batch_u8_device.argtypes = [vp, vp, sz, sz, i, i, C.c_double, C.c_double, vp, vp, sz, sz]
    L.dfx_calc_batch_u8_device.restype = i
    dp = C.POINTER(C.c_double)
    L.dfx_calc_batch_png.argtypes = [vp, C.POINTER(vp), sz, i, i, C.POINTER(vp), C.POINTER(vp), sz, dp]
    L.dfx_calc_batch_png.restype = i
    L.dfx_submit_batch_png.argtypes = [vp, C.POINTER(vp), sz, i, i, C.POINTER(vp), C.POINTER(vp), sz, dp, C.POINTER(C.c_uint64)]
    L.dfx_submit_batch_png.restype = i
    L.dfx_calc_batch_png_device.argtypes = [vp, vp, sz, sz, i, i, vp, vp, sz, sz, vp]
    L.dfx_calc_batch_png_device.restype = i
    L.dfx_flow_to_png_device.argtypes = [vp, vp, sz, i, vp, vp, sz, sz, vp]
    L.dfx_flow_to_png_device.restype = i
    L.dfx_flow_to_u8_device.argtypes = [vp, vp, sz, i, C.c_double, C.c_double, vp, vp, sz, sz]
    L.dfx_flow_to_u8_device.restype = i
    L.dfx_set_source_format.argtypes = [vp, i, i, i]
    L.dfx_set_source_format.restype = i
    L.dfx_prepare_frames.argtypes = [vp, C.POINTER(vp), sz, i, i, i, i, C.POINTER(vp), sz]
    L.dfx_prepare_frames.restype = i
    L.dfx_prepare_frames_device.argtypes = [vp, vp, sz, sz, i, i, i, i, vp, sz, sz]
    L.dfx_prepare_frames_device.restype = i
    L.dfx_get_stats.argtypes = [vp, C.POINTER(DfxStats)]
    L.dfx_get_stats.restype = i
    L.dfx_reset_stats.argtypes = [vp]
    L.dfx_last_error.argtypes = [vp]
    L.dfx_last_error.restype = C.c_char_p
    L.dfx_destroy.argtypes = [vp]
    L.dfx_device_malloc.argtypes = [vp, C.POINTER(vp), sz]
    L.dfx_device_malloc.restype = i
    L.dfx_device_free.argtypes = [vp, vp]
    L.dfx_device_free.restype = i
    L.dfx_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.dfx_memcpy_h2d.restype = i
    L.dfx_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.dfx_memcpy_d2h.restype = i
    L.dfx_host_alloc.argtypes = [C.POINTER(vp), sz]
    L.dfx_host_alloc.restype = i
    L.dfx_host_free.argtypes = [vp]
    L.dfx_host_free.restype = i
    _lib = L
    return L


def device_count() -> int:
    return int(load_library().dfx_device_count())


def algo_from_name(name: str) -> int:
    """Map -a=<name>; raises with the reference's runtime_error texts (src/denseflow_gpu.cpp:296, :336)."""
    L = load_library()
    out = C.c_int(0)
    rc = L.dfx_algo_from_name(name.encode(), C.byref(out))
    if rc != OK:
        buf = C.create_string_buffer(256)
        L.dfx_algo_error_message(rc, name.encode(), buf, 256)
        raise DfxError(rc, buf.value.decode())
    return out.value


def default_params() -> DfxParams:
    p = DfxParams()
    load_library().dfx_default_params(C.byref(p))
    return p


class FlowEngine:
    """One handle = one device + one private stream set (not thread-safe), sized for width x height."""

    def __init__(self, width: int, height: int, algorithm: str = "tvl1", device: int = 0,
                 params: DfxParams | None = None, **knobs):
        self._L = load_library()
        self._h = C.c_void_p()
        self.width, self.height = int(width), int(height)
        self.algorithm = algorithm
        algo = algo_from_name(algorithm)
        if knobs:
            if params is None:
                params = default_params()
            for k, v in knobs.items():
                setattr(params, k, v)
        rc = self._L.dfx_create(C.byref(self._h), int(device), algo, self.width, self.height,
                                C.byref(params) if params is not None else None)
        if rc != OK:
            msg = self._L.dfx_last_error(None).decode()
            self._h = C.c_void_p()
            raise DfxError(rc, msg or f"dfx_create failed with status {rc}")

    # -- helpers -----------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != OK:
            raise DfxError(rc, self._L.dfx_last_error(self._h).decode() or f"dfx status {rc}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.dfx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- frame preparation on the device (reference: cvtColor + cv::resize in load_frames_batch) ----------
    def set_source_format(self, src_width: int = 0, src_height: int = 0, channels: int = 1):
        """Frames passed to calc / calc_optflows* are src_width x src_height with 1 (gray) or 3 (BGR) channels
        from now on and are converted / resized to the engine's size on the device.  () restores the default."""
        self._check(self._L.dfx_set_source_format(self._h, int(src_width), int(src_height), int(channels)))
        self._src = (int(src_height), int(src_width)) + ((3,) if channels == 3 else ()) if src_width else None

    def _frame_shape(self):
        return getattr(self, "_src", None) or (self.height, self.width)

    def prepare_frames(self, frames):
        """cvtColor(BGR2GRAY) + cv::resize to the engine's size for a list of (h, w) or (h, w, 3) uint8 frames."""
        src = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        n = len(src)
        out = [np.empty((self.height, self.width), np.uint8) for _ in range(n)]
        if n == 0:
            return out
        shp = src[0].shape
        if any(f.shape != shp for f in src) or len(shp) not in (2, 3) or (len(shp) == 3 and shp[2] != 3):
            raise ValueError("frames must share one (h, w) or (h, w, 3) shape")
        ch = 1 if len(shp) == 2 else 3
        sp = (C.c_void_p * n)(*[f.ctypes.data for f in src])
        op = (C.c_void_p * n)(*[f.ctypes.data for f in out])
        self._check(self._L.dfx_prepare_frames(self._h, sp, shp[1] * ch, shp[1], shp[0], ch, n, op, self.width))
        return out

    def prepare_frames_device(self, d_src_ptr: int, src_pitch: int, src_frame_stride: int, src_width: int,
                              src_height: int, channels: int, n: int, d_gray_ptr: int, gray_pitch: int,
                              gray_frame_stride: int):
        self._check(self._L.dfx_prepare_frames_device(self._h, d_src_ptr, src_pitch, src_frame_stride, int(src_width),
                                                      int(src_height), int(channels), int(n), d_gray_ptr, gray_pitch,
                                                      gray_frame_stride))

    # -- several short clips in one FlowBuffer (dfx_next_segments) ------------------------------------------
    def next_segments(self, seg_frames):
        """The NEXT calc_optflows* / submit_optflows call carries len(seg_frames) clips back to back, clip s being
        seg_frames[s] consecutive frames; pairs are formed inside each clip only (outputs in clip order)."""
        self._pending_seg = [int(x) for x in seg_frames]

    def _num_pairs(self, n: int, step: int) -> int:
        seg, self._pending_seg = getattr(self, "_pending_seg", None), None
        self._armed_seg = None
        if seg is None:
            return max(n - abs(step), 0)
        if sum(seg) != n or min(seg, default=0) < 0:
            raise ValueError("segment lengths must be >= 0 and add up to the number of frames")
        self._armed_seg = seg
        return sum(max(x - abs(step), 0) for x in seg)

    def _arm(self):  # right in front of the library call the declaration is meant for
        seg, self._armed_seg = getattr(self, "_armed_seg", None), None
        if seg is not None:
            self._check(self._L.dfx_next_segments(self._h, (C.c_int * max(len(seg), 1))(*seg), len(seg)))

    # -- the hot path ------------------------------------------------------------------------
    def calc(self, frame_a: np.ndarray, frame_b: np.ndarray) -> np.ndarray:
        """alg->calc(a, b): one (H, W, 2) float32 flow, channel 0 = u (x), 1 = v (y)."""
        a = np.ascontiguousarray(frame_a, dtype=np.uint8)
        b = np.ascontiguousarray(frame_b, dtype=np.uint8)
        if a.shape != self._frame_shape() or b.shape != a.shape:
            raise ValueError("frame shape does not match the engine")
        out = np.empty((self.height, self.width, 2), dtype=np.float32)
        self._check(self._L.dfx_calc(self._h, a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0],
                                     out.ctypes.data, out.strides[0]))
        return out

    def calc_optflows(self, frames_gray, step: int):
        """The loop of DenseFlow::calc_optflows_imp (src/denseflow_gpu.cpp:307-342) for one FlowBuffer."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames_gray]
        n = len(frames)
        m = self._num_pairs(n, step)
        flows = [np.empty((self.height, self.width, 2), dtype=np.float32) for _ in range(m)]
        if m == 0:
            return flows
        for f in frames:
            if f.shape != self._frame_shape():
                raise ValueError("frame shape does not match the engine")
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        op = (C.c_void_p * m)(*[f.ctypes.data for f in flows])
        self._arm()
        self._check(self._L.dfx_calc_batch(self._h, fp, frames[0].strides[0], n, int(step), op, self.width * 8))
        return flows

    # -- asynchronous FlowBuffers (dfx_submit_batch* / dfx_wait) -----------------------------------------------
    def submit_optflows(self, frames_gray, step: int, bound: float | None = None):
        """dfx_submit_batch (bound None: float flows) or dfx_submit_batch_u8 (planes bounded to [-bound, bound]).
        Returns (ticket, outputs); the outputs are valid after wait(ticket).  The output arrays are created here and
        must be kept alive by the caller until then."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames_gray]
        n = len(frames)
        m = self._num_pairs(n, step)
        for f in frames:
            if f.shape != self._frame_shape():
                raise ValueError("frame shape does not match the engine")
        t = C.c_uint64(0)
        fp = (C.c_void_p * max(n, 1))(*[f.ctypes.data for f in frames])
        pitch = frames[0].strides[0] if n else self.width
        if bound is None:
            flows = [np.empty((self.height, self.width, 2), dtype=np.float32) for _ in range(m)]
            op = (C.c_void_p * max(m, 1))(*[f.ctypes.data for f in flows])
            self._arm()
            self._check(self._L.dfx_submit_batch(self._h, fp, pitch, n, int(step), op, self.width * 8, C.byref(t)))
            return t.value, flows
        img_x = [np.empty((self.height, self.width), dtype=np.uint8) for _ in range(m)]
        img_y = [np.empty((self.height, self.width), dtype=np.uint8) for _ in range(m)]
        xp = (C.c_void_p * max(m, 1))(*[f.ctypes.data for f in img_x])
        yp = (C.c_void_p * max(m, 1))(*[f.ctypes.data for f in img_y])
        self._arm()
        self._check(self._L.dfx_submit_batch_u8(self._h, fp, pitch, n, int(step), -float(bound), float(bound), xp, yp,
                                                self.width, C.byref(t)))
        return t.value, (img_x, img_y)

    def wait(self, ticket: int = 0):
        self._check(self._L.dfx_wait(self._h, int(ticket)))

    def calc_optflows_device(self, d_frames_ptr: int, pitch: int, frame_stride: int, n_frames: int, step: int,
                             d_flows_ptr: int, flow_stride_floats: int):
        """Frames and flows already resident in HBM (raw device pointers, e.g. torch .data_ptr())."""
        self._num_pairs(n_frames, step)
        self._arm()
        self._check(self._L.dfx_calc_batch_device(self._h, d_frames_ptr, pitch, frame_stride, n_frames, int(step),
                                                  d_flows_ptr, flow_stride_floats))

    # -- flow bounding on the device (reference: convertFlowToImage, src/common.cpp:4-16) ------
    def calc_optflows_u8(self, frames_gray, step: int, bound: float, lower: float | None = None):
        """calc_optflows followed by encodeFlowMap's bounding (src/common.cpp:48-64), all on the device.

        Returns (img_x, img_y): two lists of M (H, W) uint8 planes.  The reference bounds to
        [-bound, bound]; pass `lower` for an asymmetric interval [lower, bound]."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames_gray]
        n = len(frames)
        m = self._num_pairs(n, step)
        img_x = [np.empty((self.height, self.width), dtype=np.uint8) for _ in range(m)]
        img_y = [np.empty((self.height, self.width), dtype=np.uint8) for _ in range(m)]
        if m == 0:
            return img_x, img_y
        for f in frames:
            if f.shape != self._frame_shape():
                raise ValueError("frame shape does not match the engine")
        lo = -float(bound) if lower is None else float(lower)
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        xp = (C.c_void_p * m)(*[f.ctypes.data for f in img_x])
        yp = (C.c_void_p * m)(*[f.ctypes.data for f in img_y])
        self._arm()
        self._check(self._L.dfx_calc_batch_u8(self._h, fp, frames[0].strides[0], n, int(step), lo, float(bound), xp,
                                              yp, self.width))
        return img_x, img_y

    # -- the -st=png scheme on the device (reference: convertFlowToPngImage, src/common.cpp:18-46) ------
    def calc_optflows_png(self, frames_gray, step: int, submit: bool = False):
        """calc_optflows followed by convertFlowToPngImage's arithmetic on the device: per flow the adaptive bounds
        (minMaxLoc, the ceil(.../4)*4 rule with its `% 8 == 0 -> += 4` step) and the two convertTo(CV_8U) planes.

        Returns (img_x, img_y, bounds): two lists of M (H, W) uint8 planes and an (M, 2) float64 array of
        (bound_x, bound_y).  png_bgr() assembles the reference's 3-channel image from them.  submit=True goes through
        dfx_submit_batch_png + dfx_wait (the host shell's form)."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames_gray]
        n = len(frames)
        m = self._num_pairs(n, step)
        img_x = [np.empty((self.height, self.width), dtype=np.uint8) for _ in range(m)]
        img_y = [np.empty((self.height, self.width), dtype=np.uint8) for _ in range(m)]
        bounds = np.zeros((m, 2), np.float64)
        if m == 0:
            return img_x, img_y, bounds
        for f in frames:
            if f.shape != self._frame_shape():
                raise ValueError("frame shape does not match the engine")
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        xp = (C.c_void_p * m)(*[f.ctypes.data for f in img_x])
        yp = (C.c_void_p * m)(*[f.ctypes.data for f in img_y])
        bp = bounds.ctypes.data_as(C.POINTER(C.c_double))
        self._arm()
        if submit:
            t = C.c_uint64(0)
            self._check(self._L.dfx_submit_batch_png(self._h, fp, frames[0].strides[0], n, int(step), xp, yp, self.width,
                                                     bp, C.byref(t)))
            self._check(self._L.dfx_wait(self._h, t.value))
        else:
            self._check(self._L.dfx_calc_batch_png(self._h, fp, frames[0].strides[0], n, int(step), xp, yp, self.width, bp))
        return img_x, img_y, bounds

    def calc_optflows_png_device(self, d_frames_ptr: int, pitch: int, frame_stride: int, n_frames: int, step: int,
                                 d_img_x_ptr: int, d_img_y_ptr: int, img_pitch: int, img_stride: int, d_bounds_ptr: int):
        self._num_pairs(n_frames, step)
        self._arm()
        self._check(self._L.dfx_calc_batch_png_device(self._h, d_frames_ptr, pitch, frame_stride, n_frames, int(step),
                                                      d_img_x_ptr, d_img_y_ptr, img_pitch, img_stride, d_bounds_ptr))

    def flow_to_png_device(self, d_flows_ptr: int, flow_stride_floats: int, n: int, d_img_x_ptr: int, d_img_y_ptr: int,
                           img_pitch: int, img_stride: int, d_bounds_ptr: int):
        self._check(self._L.dfx_flow_to_png_device(self._h, d_flows_ptr, flow_stride_floats, n, d_img_x_ptr, d_img_y_ptr,
                                                   img_pitch, img_stride, d_bounds_ptr))

    @staticmethod
    def png_bgr(img_x: np.ndarray, img_y: np.ndarray, bound_x: float, bound_y: float) -> np.ndarray:
        """The reference's 3-channel PNG image (src/common.cpp:41-45) from the device's planes and bounds: channel 2 is
        bound_x / 4 on rows 0 .. int(H / 2) (rectangle's inclusive box, Point's truncation) and bound_y / 4 below."""
        h, w = img_x.shape
        out = np.empty((h, w, 3), np.uint8)
        out[..., 0], out[..., 1] = img_x, img_y
        half = int(h / 2)
        sat = lambda v: int(min(255, max(0, np.rint(v))))  # noqa: E731 — saturate_cast<uchar>(double): cvRound, clamp
        out[: half + 1, :, 2] = sat(bound_x / 4)
        out[half + 1:, :, 2] = sat(bound_y / 4)
        return out

    def calc_optflows_jpeg(self, frames_gray, step: int, bound: float, quality: int = 95):
        """encodeFlowMap of every flow of the FlowBuffer on the device (src/common.cpp:48-64): returns two lists of
        `bytes`, the flow_x and flow_y JPEG files (bounded to [-bound, bound], quality like cv::imencode)."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames_gray]
        n = len(frames)
        m = self._num_pairs(n, step)
        if m == 0:
            return [], []
        for f in frames:
            if f.shape != self._frame_shape():
                raise ValueError("frame shape does not match the engine")
        cap = int(self._L.dfx_jpeg_capacity(self._h))
        bx = [np.empty(cap, np.uint8) for _ in range(m)]
        by = [np.empty(cap, np.uint8) for _ in range(m)]
        sx, sy = (C.c_uint32 * m)(), (C.c_uint32 * m)()
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        self._arm()
        self._check(self._L.dfx_calc_batch_jpeg(self._h, fp, frames[0].strides[0], n, int(step), -float(bound),
                                                float(bound), int(quality), (C.c_void_p * m)(*[b.ctypes.data for b in bx]),
                                                (C.c_void_p * m)(*[b.ctypes.data for b in by]), cap, sx, sy))
        return ([bx[i][:sx[i]].tobytes() for i in range(m)], [by[i][:sy[i]].tobytes() for i in range(m)])

    def encode_jpeg(self, planes, quality: int = 95):
        """imencode(".jpg") of (H, W) uint8 planes on the device: a list of `bytes`."""
        ps = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        n = len(ps)
        if n == 0:
            return []
        if any(p.shape != (self.height, self.width) for p in ps):
            raise ValueError("plane shape does not match the engine")
        cap = int(self._L.dfx_jpeg_capacity(self._h))
        bufs = [np.empty(cap, np.uint8) for _ in range(n)]
        sizes = (C.c_uint32 * n)()
        self._check(self._L.dfx_encode_jpeg(self._h, (C.c_void_p * n)(*[p.ctypes.data for p in ps]), self.width, n,
                                            int(quality), (C.c_void_p * n)(*[b.ctypes.data for b in bufs]), cap, sizes))
        return [bufs[i][:sizes[i]].tobytes() for i in range(n)]

    def calc_optflows_u8_device(self, d_frames_ptr: int, pitch: int, frame_stride: int, n_frames: int, step: int,
                                lower: float, upper: float, d_img_x_ptr: int, d_img_y_ptr: int, img_pitch: int,
                                img_stride: int):
        """Frames and bounded planes resident in HBM (raw device pointers)."""
        self._check(self._L.dfx_calc_batch_u8_device(self._h, d_frames_ptr, pitch, frame_stride, n_frames, int(step),
                                                     float(lower), float(upper), d_img_x_ptr, d_img_y_ptr,
                                                     img_pitch, img_stride))

    def flow_to_u8_device(self, d_flows_ptr: int, flow_stride_floats: int, n: int, lower: float, upper: float,
                          d_img_x_ptr: int, d_img_y_ptr: int, img_pitch: int, img_stride: int):
        """Bound n flows that are already in device memory."""
        self._check(self._L.dfx_flow_to_u8_device(self._h, d_flows_ptr, flow_stride_floats, int(n), float(lower),
                                                  float(upper), d_img_x_ptr, d_img_y_ptr, img_pitch, img_stride))

    def stats(self) -> DfxStats:
        s = DfxStats()
        self._check(self._L.dfx_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._L.dfx_reset_stats(self._h)
