"""TEST INFRASTRUCTURE: what cv::imencode(".png") writes, produced by the real libpng.

The reference's encodeFlowMapPng ends in imencode(".png", bgr) (/root/reference/src/common.cpp:70).  OpenCV's PngEncoder
(modules/imgcodecs/src/grfmt_png.cpp, 4.5.2) makes these libpng calls for an 8-bit image and default parameters:

    png_create_write_struct / png_create_info_struct / png_set_write_fn (in-memory sink)
    png_set_filter(PNG_FILTER_TYPE_BASE, PNG_FILTER_SUB); png_set_compression_level(Z_BEST_SPEED)   ("tune for speed")
    png_set_compression_strategy(IMWRITE_PNG_STRATEGY_RLE = Z_RLE)
    png_set_IHDR(w, h, 8, GRAY | RGB, PNG_INTERLACE_NONE, default, default); png_write_info
    png_set_bgr; png_write_image(rows); png_write_end

libpng is not in /root/reference, but the system's libpng16.so.16 (and its zlib) is loadable here and on the GPU box;
this module drives it through ctypes with exactly that sequence."""
import ctypes as C
import ctypes.util

import numpy as np

PNG_COLOR_TYPE_GRAY, PNG_COLOR_TYPE_RGB = 0, 2
PNG_FILTER_SUB = 0x10
Z_BEST_SPEED, Z_RLE = 1, 3


def load():
    for name in ("libpng16.so.16", ctypes.util.find_library("png16") or "", ctypes.util.find_library("png") or ""):
        if not name:
            continue
        try:
            L = C.CDLL(name)
        except OSError:
            continue
        L.png_get_libpng_ver.restype = C.c_char_p
        L.png_create_write_struct.restype = C.c_void_p
        L.png_create_write_struct.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.png_create_info_struct.restype = C.c_void_p
        L.png_create_info_struct.argtypes = [C.c_void_p]
        return L
    return None


_WRITE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_ubyte), C.c_size_t)
_FLUSH = C.CFUNCTYPE(None, C.c_void_p)


def imencode_png(img: np.ndarray) -> bytes:
    """img: (H, W) uint8 gray or (H, W, 3) uint8 BGR."""
    L = load()
    assert L is not None, "no libpng16"
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else 3
    out = bytearray()

    def sink(_png, data, n):
        out.extend(C.string_at(data, n))

    wfn, ffn = _WRITE(sink), _FLUSH(lambda _png: None)
    png = C.c_void_p(L.png_create_write_struct(L.png_get_libpng_ver(None), None, None, None))
    info = C.c_void_p(L.png_create_info_struct(png))
    assert png and info
    L.png_set_write_fn(png, None, wfn, ffn)
    L.png_set_filter(png, 0, PNG_FILTER_SUB)
    L.png_set_compression_level(png, Z_BEST_SPEED)
    L.png_set_compression_strategy(png, Z_RLE)
    L.png_set_IHDR(png, info, C.c_uint32(w), C.c_uint32(h), 8, PNG_COLOR_TYPE_GRAY if ch == 1 else PNG_COLOR_TYPE_RGB, 0, 0, 0)
    L.png_write_info(png, info)
    L.png_set_bgr(png)
    rows = (C.c_void_p * h)(*[img.ctypes.data + y * w * ch for y in range(h)])
    L.png_write_image(png, rows)
    L.png_write_end(png, info)
    pp, ip = C.c_void_p(png.value), C.c_void_p(info.value)
    L.png_destroy_write_struct(C.byref(pp), C.byref(ip))
    return bytes(out)
