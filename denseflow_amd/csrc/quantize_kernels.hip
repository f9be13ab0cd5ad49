// quantize_kernels.hip — flow bounding on the device.
//
// Replaces the scalar loop of convertFlowToImage (/root/reference/src/common.cpp:4-16) that the
// reference's encode thread runs on every flow before JPEG encoding (encodeFlowMap :48-64, called
// from DenseFlow::encode_save src/denseflow_gpu.cpp:396-454):
//     CAST(v, L, H) = v > H ? 255 : v < L ? 0 : cvRound(255 * (v - L) / (H - L))
// evaluated in double with cvRound = round-half-to-even, v the float flow component.  Doing it here
// means 2 bytes per pixel cross PCIe instead of 8 and the host encoder starts from finished planes.
//
// HBM-bound streaming kernel: 8 B read + 2 B written per pixel, 4 pixels per lane (two 16-byte loads,
// two 4-byte stores per lane -> full 64-lane coalescing).
#include "quantize_kernels.h"

#include <algorithm>

namespace {

__device__ __forceinline__ unsigned quant_cast(float vf, double lo, double hi, double span) {
    const double v = (double)vf;
    // IEEE double subtract, multiply and divide in this order; v_rndne_f64 rounds half to even.
    const double q = __builtin_rint(255.0 * (v - lo) / span);
    // NaN: both comparisons are false and cvRound(NaN) is INT_MIN (cvtsd2si), whose low byte is 0.
    const unsigned r = (q == q) ? (unsigned)(int)q & 0xffu : 0u;
    return v > hi ? 255u : (v < lo ? 0u : r);
}

__global__ __launch_bounds__(256) void k_flow_to_u8(const float *flows, long long flow_stride, int w, int h,
                                                     double lo, double hi, unsigned char *img_x,
                                                     unsigned char *img_y, long long img_pitch,
                                                     long long img_stride) {
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int pair = blockIdx.z;
    if (x4 >= w || y >= h)
        return;
    const double span = hi - lo;
    const float *src = flows + (size_t)pair * flow_stride + ((size_t)y * w + x4) * 2;
    unsigned char *dx = img_x + (size_t)pair * img_stride + (size_t)y * img_pitch + x4;
    unsigned char *dy = img_y + (size_t)pair * img_stride + (size_t)y * img_pitch + x4;
    // rows of the dense flow start at y*w*8 bytes: 16-byte alignment needs (y*w + x4) even -> always true for
    // x4 % 4 == 0 when w is even; fall back to scalar accesses otherwise and at the ragged right edge.
    const bool wide = (x4 + 4 <= w) && ((((size_t)src) & 15) == 0) && ((((size_t)dx | (size_t)dy) & 3) == 0);
    if (wide) {
        const float4 a = *reinterpret_cast<const float4 *>(src);
        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
        const unsigned px = quant_cast(a.x, lo, hi, span) | (quant_cast(a.z, lo, hi, span) << 8) |
                            (quant_cast(b.x, lo, hi, span) << 16) | (quant_cast(b.z, lo, hi, span) << 24);
        const unsigned py = quant_cast(a.y, lo, hi, span) | (quant_cast(a.w, lo, hi, span) << 8) |
                            (quant_cast(b.y, lo, hi, span) << 16) | (quant_cast(b.w, lo, hi, span) << 24);
        *reinterpret_cast<unsigned *>(dx) = px;
        *reinterpret_cast<unsigned *>(dy) = py;
    } else {
        for (int i = 0; i < 4 && x4 + i < w; ++i) {
            dx[i] = (unsigned char)quant_cast(src[2 * i], lo, hi, span);
            dy[i] = (unsigned char)quant_cast(src[2 * i + 1], lo, hi, span);
        }
    }
}

// ---- the -st=png scheme: convertFlowToPngImage, /root/reference/src/common.cpp:18-46 ------------------------------
// Per flow: minMaxLoc of u and of v; bound_x = min(1020, ceil((min(w, max|u|) * 128 / 127) / 4) * 4), + 4 when that
// integer is a multiple of 8 (bound_y: h, v); the two convertTo(CV_8U, 1 / (bound / 128), 128) planes; the third
// channel (bound / 4, x's value on rows 0 .. int(h / 2), y's below) is two bytes the host writes from the bounds.
// 2 B/px + 16 B per flow leave the device instead of 8 B/px, and the host encoder starts from finished planes.

// float -> unsigned key whose integer order is the float order (-0 < +0, which abs() below makes irrelevant)
__device__ __forceinline__ unsigned png_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float png_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// mm[4 * flow + {0: min u, 1: max u, 2: min v, 3: max v}] as keys; (re)armed for n flows
__global__ void k_png_minmax_init(unsigned *mm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * n)
        mm[i] = (i & 1) ? 0u : 0xffffffffu;
}

// Exact and order-independent (min / max of floats), so atomics keep the result deterministic.  NaNs never win a
// comparison, as in minMaxLoc.  One float4 (two pixels) per lane and trip, grid-stride over the flow.
__global__ __launch_bounds__(256) void k_png_minmax(const float *flows, long long flow_stride, long long n_px,
                                                     unsigned *mm) {
    const int flow = blockIdx.z;
    const float *src = flows + (size_t)flow * flow_stride;
    float lo_u = INFINITY, hi_u = -INFINITY, lo_v = INFINITY, hi_v = -INFINITY;
    const long long n4 = n_px / 2; // float4 = (u, v, u, v); flows are dense and 16-byte aligned when W*H is even ...
    const bool wide = ((((size_t)src) & 15) == 0);
    const long long stride = (long long)gridDim.x * 256;
    if (wide) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const float4 a = reinterpret_cast<const float4 *>(src)[i];
            lo_u = fminf(lo_u, fminf(a.x, a.z)), hi_u = fmaxf(hi_u, fmaxf(a.x, a.z));
            lo_v = fminf(lo_v, fminf(a.y, a.w)), hi_v = fmaxf(hi_v, fmaxf(a.y, a.w));
        }
    }
    // ... the odd last pixel (and everything, when the flow is not 16-byte aligned) goes pixel by pixel
    for (long long i = (wide ? 2 * n4 : 0) + (long long)blockIdx.x * 256 + threadIdx.x; i < n_px; i += stride) {
        const float u = src[2 * i], v = src[2 * i + 1];
        lo_u = fminf(lo_u, u), hi_u = fmaxf(hi_u, u);
        lo_v = fminf(lo_v, v), hi_v = fmaxf(hi_v, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo_u = fminf(lo_u, __shfl_xor(lo_u, off, 64)), hi_u = fmaxf(hi_u, __shfl_xor(hi_u, off, 64));
        lo_v = fminf(lo_v, __shfl_xor(lo_v, off, 64)), hi_v = fmaxf(hi_v, __shfl_xor(hi_v, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned *m = mm + 4 * flow;
        if (lo_u <= hi_u) { // at least one non-NaN value seen by this wave
            atomicMin(m + 0, png_key(lo_u));
            atomicMax(m + 1, png_key(hi_u));
        }
        if (lo_v <= hi_v) {
            atomicMin(m + 2, png_key(lo_v));
            atomicMax(m + 3, png_key(hi_v));
        }
    }
}

struct PngScale {
    float ax, ay; // eps_x_inv, eps_y_inv
};

// src/common.cpp:24-35, in double, operation for operation (IEEE divide, ceil; no contraction in this library)
__device__ __forceinline__ double png_bound(double extent, float mn, float mx) {
    const double a = fabs((double)mn), b = fabs((double)mx);
    const double m = a > b ? a : b;
    const double c = extent < m ? extent : m;
    double bound = ceil((c * 128. / 127.) / 4) * 4;
    bound = bound > 255. * 4 ? 255. * 4 : bound;
    if ((int)bound % 8 == 0)
        bound += 4;
    return bound;
}

// one thread per flow: bounds[2 * flow] = {bound_x, bound_y}, scale[flow] = the two float factors of convertTo
__global__ void k_png_bounds(const unsigned *mm, int n, int w, int h, double *bounds, PngScale *scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double base = 1. / 128.;
    // a plane without a single comparable value (NaNs only) leaves its keys as armed: minMaxLoc reports 0 / 0 then
    const bool none_u = mm[4 * i + 0] == 0xffffffffu && mm[4 * i + 1] == 0u;
    const bool none_v = mm[4 * i + 2] == 0xffffffffu && mm[4 * i + 3] == 0u;
    const double bx = png_bound((double)w, none_u ? 0.0f : png_unkey(mm[4 * i + 0]), none_u ? 0.0f : png_unkey(mm[4 * i + 1]));
    const double by = png_bound((double)h, none_v ? 0.0f : png_unkey(mm[4 * i + 2]), none_v ? 0.0f : png_unkey(mm[4 * i + 3]));
    bounds[2 * i] = bx;
    bounds[2 * i + 1] = by;
    PngScale s;
    s.ax = (float)(1. / (base * bx));
    s.ay = (float)(1. / (base * by));
    scale[i] = s;
}

// Mat::convertTo(CV_8U, alpha, 128) on a float source: saturate_cast<uchar>(v * (float)alpha + 128.f), product and sum
// rounded separately in float, cvRound (half to even), clamped.  NaN -> cvRound = INT_MIN -> 0.
__device__ __forceinline__ unsigned png_cast(float v, float a) {
    const float p = v * a;
    const float t = p + 128.f;
    const float r = __builtin_rintf(t);
    return (t == t) ? (unsigned)(int)fminf(fmaxf(r, 0.f), 255.f) : 0u;
}

__global__ __launch_bounds__(256) void k_flow_to_png_planes(const float *flows, long long flow_stride, int w, int h,
                                                             const PngScale *scale, unsigned char *img_x,
                                                             unsigned char *img_y, long long img_pitch,
                                                             long long img_stride) {
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int pair = blockIdx.z;
    if (x4 >= w || y >= h)
        return;
    const PngScale sc = scale[pair];
    const float *src = flows + (size_t)pair * flow_stride + ((size_t)y * w + x4) * 2;
    unsigned char *dx = img_x + (size_t)pair * img_stride + (size_t)y * img_pitch + x4;
    unsigned char *dy = img_y + (size_t)pair * img_stride + (size_t)y * img_pitch + x4;
    const bool wide = (x4 + 4 <= w) && ((((size_t)src) & 15) == 0) && ((((size_t)dx | (size_t)dy) & 3) == 0);
    if (wide) {
        const float4 a = *reinterpret_cast<const float4 *>(src);
        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
        *reinterpret_cast<unsigned *>(dx) = png_cast(a.x, sc.ax) | (png_cast(a.z, sc.ax) << 8) |
                                            (png_cast(b.x, sc.ax) << 16) | (png_cast(b.z, sc.ax) << 24);
        *reinterpret_cast<unsigned *>(dy) = png_cast(a.y, sc.ay) | (png_cast(a.w, sc.ay) << 8) |
                                            (png_cast(b.y, sc.ay) << 16) | (png_cast(b.w, sc.ay) << 24);
    } else {
        for (int i = 0; i < 4 && x4 + i < w; ++i) {
            dx[i] = (unsigned char)png_cast(src[2 * i], sc.ax);
            dy[i] = (unsigned char)png_cast(src[2 * i + 1], sc.ay);
        }
    }
}

} // namespace

size_t quant_png_scratch_bytes(int n) { return (size_t)n * (4 * sizeof(unsigned) + sizeof(PngScale)); }

void quant_launch_flow_to_png_planes(hipStream_t s, const float *d_flows, long long flow_stride, int n, int w, int h,
                                     void *d_scratch, double *d_bounds, unsigned char *d_img_x, unsigned char *d_img_y,
                                     long long img_pitch, long long img_stride) {
    if (n <= 0)
        return;
    unsigned *mm = reinterpret_cast<unsigned *>(d_scratch);
    PngScale *scale = reinterpret_cast<PngScale *>(mm + 4 * (size_t)n);
    const long long n_px = (long long)w * h;
    hipLaunchKernelGGL(k_png_minmax_init, dim3((4 * n + 255) / 256), dim3(256), 0, s, mm, n);
    // enough workgroups per flow to fill the device at small batches, few enough that the atomics do not matter
    const int per_flow = (int)std::max<long long>(1, std::min<long long>(256, n_px / (256 * 16)));
    hipLaunchKernelGGL(k_png_minmax, dim3(per_flow, 1, n), dim3(256), 0, s, d_flows, flow_stride, n_px, mm);
    hipLaunchKernelGGL(k_png_bounds, dim3((n + 63) / 64), dim3(64), 0, s, mm, n, w, h, d_bounds, scale);
    const dim3 grid((w + 255) / 256, (h + 3) / 4, n);
    hipLaunchKernelGGL(k_flow_to_png_planes, grid, dim3(256), 0, s, d_flows, flow_stride, w, h, scale, d_img_x, d_img_y,
                       img_pitch, img_stride);
}

void quant_launch_flow_to_u8(hipStream_t s, const float *d_flows, long long flow_stride, int n, int w, int h,
                             double lo, double hi, unsigned char *d_img_x, unsigned char *d_img_y,
                             long long img_pitch, long long img_stride) {
    if (n <= 0)
        return;
    const dim3 grid((w + 255) / 256, (h + 3) / 4, n);
    hipLaunchKernelGGL(k_flow_to_u8, grid, dim3(256), 0, s, d_flows, flow_stride, w, h, lo, hi, d_img_x, d_img_y,
                       img_pitch, img_stride);
}
