// farneback_kernels.hip — hand-written gfx950 kernels for the -a=farn hot path.
//
// Replaces the CUDA kernels cv::cuda::FarnebackOpticalFlow::calc launches for the reference
// (/root/reference/src/denseflow_gpu.cpp:329): gaussianBlur, resize, polynomialExpansion<5>,
// updateMatrices, boxFilter5, updateFlow, convertTo, multiply, merge.
//
// What is restructured for MI355X (same arithmetic, same accumulation order, far fewer bytes):
//   * upstream blurs the FULL-resolution frame at every pyramid level (up to 79 taps) and then
//     bilinear-resizes it; here the separable blur is evaluated only at the rows/columns the resize
//     samples (blur_v at 2 rows per destination row, blur_h at 2 columns per destination pixel);
//   * boxFilter5 + updateFlow + updateMatrices are one launch per iteration: the 13x13 box sums are
//     built in LDS from a halo tile, the 2x2 solve and the next M are computed in registers, so the
//     box-filtered M is never written to HBM;
//   * a frame's polynomial expansion is computed once and serves two pairs.
// Compiled with -ffp-contract=off: every a*b+c is a rounded multiply then a rounded add, like the oracle.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <algorithm>
#include <cstdlib>

#include "dfx_device.h"
#include "farneback_kernels.h"
#include "farneback_plan.h"

#define FARN_HALF_MAX 8 // box half-width supported by the fused iteration kernel (winSize <= 17)

static inline dim3 grid64x4(int w, int h, int z) { return dim3((w + 63) / 64, (h + 3) / 4, z); }

// ------------------------------------------------------------------------------------------------
// helpers

__device__ __forceinline__ int reflect101_low(int x, int last) { return abs(x) % (last + 1); }
__device__ __forceinline__ int reflect101_high(int x, int last) { return abs(last - abs(last - x)) % (last + 1); }
__device__ __forceinline__ int reflect101(int x, int last) { return reflect101_low(reflect101_high(x, last), last); }

__device__ __forceinline__ float *farn_plane(const FarnPairCtx &c, int pair, int plane) {
    return c.planes + (long long)pair * c.slot_stride + (long long)plane * c.plane_stride;
}

// ------------------------------------------------------------------------------------------------
// frame preparation

__global__ __launch_bounds__(256) void k_farn_u8_to_f32(const unsigned char *src, long long src_frame_stride,
                                                        long long src_pitch, float *dst, long long dst_frame_stride,
                                                        int w, int h, int pitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    const int z = blockIdx.z;
    dst[(long long)z * dst_frame_stride + (long long)y * pitch + x] =
        (float)src[(long long)z * src_frame_stride + (long long)y * src_pitch + x];
}

// tmpv[z][2*dy + r][x] = vertical Gaussian pass (B.4) of frame z at source row (r ? y2r : y1r) of
// destination row dy (E.1 row mapping), all full-resolution columns x.
// SRC = unsigned char: the frames as the caller handed them over (E.2's convertTo(CV_32F) is the (float) of each load —
// exact, so the separate u8 -> f32 pass and its plane are not needed); SRC = float: a converted plane (spitch = pitch0).
template <class SRC>
__global__ __launch_bounds__(256) void k_farn_blur_v(const SRC *__restrict__ frames, long long frame_stride, long long spitch,
                                                     int W, int H, int pitch0, int dst_h, float ify,
                                                     const float *__restrict__ ker, int half,
                                                     float *__restrict__ tmpv, long long tmpv_frame_stride,
                                                     int skip_zero_weights) {
    // one thread = one column of one destination row: BOTH source rows the bilinear resize samples (y1, y1 + 1).
    // Their tap windows overlap in all but one row each, so the interior path loads 2 values per tap pair for the
    // two outputs (a rolling pair of registers supplies the other two) instead of 4.
    const DfxBlockXY blk = dfx_block_xy(); // the tap windows of neighbouring rows overlap: keep them in one L2
    const int x = blk.x * 64 + (threadIdx.x & 63);
    const int dy = blk.y * 4 + (threadIdx.x >> 6);
    if (x >= W || dy >= dst_h)
        return;
    const float sy = (float)dy * ify;
    const int y1 = (int)floorf(sy);
    const int yr0 = min(y1, H - 1), yr1 = min(y1 + 1, H - 1);
    const SRC *src = frames + (long long)blockIdx.z * frame_stride;
    float *out = tmpv + (long long)blockIdx.z * tmpv_frame_stride + (long long)(2 * dy) * pitch0 + x;
    // The resize weighs source row y1 + 1 with (sy - y1): exactly 0 whenever the level's scale divides the frame
    // (every level of a 2^k pyramid but the odd-sized ones).  k_farn_blur_h_resize then never reads that row, so it
    // is not computed either (skip_zero_weights: A/B switch, same bits both ways).
    if (skip_zero_weights && sy - (float)y1 == 0.0f) {
        float v;
        if (yr0 - half >= 0 && yr0 + half <= H - 1) {
            const SRC *c0 = src + (long long)yr0 * spitch + x;
            v = (float)c0[0] * ker[0];
#pragma unroll 8
            for (int j = 1; j <= half; ++j)
                v = v + ((float)c0[-(long long)j * spitch] + (float)c0[(long long)j * spitch]) * ker[j];
        } else {
            v = (float)src[(long long)yr0 * spitch + x] * ker[0];
            for (int j = 1; j <= half; ++j) {
                const float a = (float)src[(long long)reflect101_low(yr0 - j, H - 1) * spitch + x];
                const float b = (float)src[(long long)reflect101_high(yr0 + j, H - 1) * spitch + x];
                v = v + (a + b) * ker[j];
            }
        }
        out[0] = v;
        return;
    }
    if (yr1 == yr0 + 1 && yr0 - half >= 0 && yr1 + half <= H - 1) {
        const SRC *c0 = src + (long long)yr0 * spitch + x, *c1 = c0 + spitch;
        float lo_prev = (float)c0[0], hi_prev = (float)c1[0]; // row yr0 - (j-1) seen from output 1, row yr1 + (j-1) from output 0
        float v0 = lo_prev * ker[0], v1 = hi_prev * ker[0];
#pragma unroll 8
        for (int j = 1; j <= half; ++j) {
            const float lo = (float)c0[-(long long)j * spitch]; // row yr0 - j
            const float hi = (float)c1[(long long)j * spitch];  // row yr1 + j
            v0 = v0 + (lo + hi_prev) * ker[j];           // rows yr0 - j and yr0 + j = yr1 + (j-1)
            v1 = v1 + (lo_prev + hi) * ker[j];           // rows yr1 - j = yr0 - (j-1) and yr1 + j
            lo_prev = lo;
            hi_prev = hi;
        }
        out[0] = v0;
        out[pitch0] = v1;
        return;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) { // frame border: BORDER_REFLECT_101 per tap
        const int yr = r ? yr1 : yr0;
        float v = (float)src[(long long)yr * spitch + x] * ker[0];
        for (int j = 1; j <= half; ++j) {
            const float a = (float)src[(long long)reflect101_low(yr - j, H - 1) * spitch + x];
            const float b = (float)src[(long long)reflect101_high(yr + j, H - 1) * spitch + x];
            v = v + (a + b) * ker[j];
        }
        out[(long long)r * pitch0] = v;
    }
}

// pyr[z][dy][dx] = bilinear (E.1) of the blurred frame, the blur's horizontal pass (B.4) being
// evaluated only at the two source columns this pixel samples.
__global__ __launch_bounds__(256) void k_farn_blur_h_resize(const float *__restrict__ tmpv,
                                                            long long tmpv_frame_stride, int W, int H, int pitch0,
                                                            int dst_w, int dst_h, int dst_pitch, float ifx, float ify,
                                                            const float *__restrict__ ker, int half,
                                                            float *__restrict__ pyr, long long pyr_frame_stride,
                                                            int skip_zero_weights) {
    const DfxBlockXY blk = dfx_block_xy(); // XCD-aware (dfx_device.h): neighbouring rows share one L2
    const int dx = blk.x * 64 + (threadIdx.x & 63);
    const int dy = blk.y * 4 + (threadIdx.x >> 6);
    if (dx >= dst_w || dy >= dst_h)
        return;
    const float sx = (float)dx * ifx, sy = (float)dy * ify;
    const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
    const int x2 = x1 + 1, y2 = y1 + 1;
    const int xc[2] = {min(x1, W - 1), min(x2, W - 1)};
    const float *rows = tmpv + (long long)blockIdx.z * tmpv_frame_stride + (long long)(2 * dy) * pitch0;
    // A bilinear tap with weight exactly 0 adds +0 to a finite, non-negative sum (blurred 8-bit frames): it is not
    // evaluated.  With a 2^k pyramid that is 3 of the 4 taps at every level whose size divides the frame's —
    // the resize degenerates to point sampling of the blurred frame, and level 0 to the blur itself.
    const bool use_k1 = !skip_zero_weights || (sx - (float)x1) != 0.0f;
    const bool use_r1 = !skip_zero_weights || (sy - (float)y1) != 0.0f;
    float bl[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float *row = rows + (long long)r * pitch0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if ((r == 1 && !use_r1) || (k == 1 && !use_k1)) {
                bl[r][k] = 0.0f;
                continue;
            }
            const int cx = xc[k];
            float res = row[cx] * ker[0];
            if (cx - half >= 0 && cx + half <= W - 1) { // window inside the row: plain indices
#pragma unroll 8
                for (int i = 1; i <= half; ++i)
                    res = res + (row[cx - i] + row[cx + i]) * ker[i];
            } else {
                for (int i = 1; i <= half; ++i)
                    res = res + (row[reflect101(cx - i, W - 1)] + row[reflect101(cx + i, W - 1)]) * ker[i];
            }
            bl[r][k] = res;
        }
    }
    (void)H;
    float out = 0.0f;
    out = out + bl[0][0] * (((float)x2 - sx) * ((float)y2 - sy));
    out = out + bl[0][1] * ((sx - (float)x1) * ((float)y2 - sy));
    out = out + bl[1][0] * (((float)x2 - sx) * (sy - (float)y1));
    out = out + bl[1][1] * ((sx - (float)x1) * (sy - (float)y1));
    pyr[(long long)blockIdx.z * pyr_frame_stride + (long long)dy * dst_pitch + dx] = out;
}

// B.5: one workgroup = one row segment of 256 - 2*5 output pixels; vertical pass into LDS, then the
// horizontal combinations.  (polyN = 5.)
__global__ __launch_bounds__(256) void k_farn_polyexp(const float *pyr, long long pyr_frame_stride,
                                                      const int *frame_slots, float *frame_R, long long frame_stride,
                                                      FarnLevelGeom L, FarnPolyConsts pc) {
    constexpr int N = 5;
    __shared__ float row[3][256];
    const int tx = threadIdx.x;
    const int y = blockIdx.y;
    const int x = blockIdx.x * (256 - 2 * N) + tx - N;
    const float *src = pyr + (long long)blockIdx.z * pyr_frame_stride;
    const int xw = min(max(x, 0), L.w - 1);
    {
        float a0 = src[(long long)y * L.pitch + xw] * pc.g[0];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            const float t0 = src[(long long)max(y - k, 0) * L.pitch + xw];
            const float t1 = src[(long long)min(y + k, L.h - 1) * L.pitch + xw];
            a0 = a0 + pc.g[k] * (t0 + t1);
            a1 = a1 + pc.xg[k] * (t1 - t0);
            a2 = a2 + pc.xxg[k] * (t0 + t1);
        }
        row[0][tx] = a0;
        row[1][tx] = a1;
        row[2][tx] = a2;
    }
    __syncthreads();
    if (tx >= N && tx + N < 256 && x < L.w) {
        float b1 = pc.g[0] * row[0][tx];
        float b3 = pc.g[0] * row[1][tx];
        float b5 = pc.g[0] * row[2][tx];
        float b2 = 0, b4 = 0, b6 = 0;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            b1 = b1 + (row[0][tx + k] + row[0][tx - k]) * pc.g[k];
            b4 = b4 + (row[0][tx + k] + row[0][tx - k]) * pc.xxg[k];
            b2 = b2 + (row[0][tx + k] - row[0][tx - k]) * pc.xg[k];
            b3 = b3 + (row[1][tx + k] + row[1][tx - k]) * pc.g[k];
            b6 = b6 + (row[1][tx + k] - row[1][tx - k]) * pc.xg[k];
            b5 = b5 + (row[2][tx + k] + row[2][tx - k]) * pc.g[k];
        }
        const long long ps = (long long)L.pitch * L.h;
        float *R = frame_R + (long long)frame_slots[blockIdx.z] * frame_stride + L.r_off + (long long)y * L.pitch + x;
        R[0] = b3 * pc.ig11;
        R[ps] = b2 * pc.ig11;
        R[2 * ps] = b1 * pc.ig03 + b5 * pc.ig33;
        R[3 * ps] = b1 * pc.ig03 + b4 * pc.ig33;
        R[4 * ps] = b6 * pc.ig55;
    }
}

// B.5 again, ROWS image rows per workgroup: a thread walks down its column with the 11 source values of the vertical
// pass in registers (one new load per row instead of eleven), the three vertical sums of a row go to one of two LDS row
// buffers (one barrier per row) and the horizontal combinations are the ones of k_farn_polyexp, in the same order.
template <int ROWS>
__global__ __launch_bounds__(256) void k_farn_polyexp_rows(const float *__restrict__ pyr, long long pyr_frame_stride,
                                                           const int *__restrict__ frame_slots,
                                                           float *__restrict__ frame_R, long long frame_stride,
                                                           FarnLevelGeom L, FarnPolyConsts pc) {
    constexpr int N = 5;
    __shared__ float row[2][3][256];
    const int tx = threadIdx.x;
    const DfxBlockXY blk = dfx_block_xy();
    const int y0 = blk.y * ROWS;
    const int x = blk.x * (256 - 2 * N) + tx - N;
    const float *src = pyr + (long long)blockIdx.z * pyr_frame_stride;
    const int xw = min(max(x, 0), L.w - 1);
    const bool writes = tx >= N && tx + N < 256 && x < L.w;
    const long long ps = (long long)L.pitch * L.h;
    float *Rbase = frame_R + (long long)frame_slots[blockIdx.z] * frame_stride + L.r_off + x;
    float win[2 * N + 1]; // win[k] = source row clamp(y - N + k)
#pragma unroll
    for (int k = 0; k <= 2 * N; ++k)
        win[k] = src[(long long)min(max(y0 - N + k, 0), L.h - 1) * L.pitch + xw];
#pragma unroll 2
    for (int r = 0; r < ROWS; ++r) {
        const int y = y0 + r;
        if (y >= L.h)
            break; // the whole workgroup
        const float nxt = src[(long long)min(y + N + 1, L.h - 1) * L.pitch + xw]; // enters the window at row y + 1
        float a0 = win[N] * pc.g[0];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            const float t0 = win[N - k];
            const float t1 = win[N + k];
            a0 = a0 + pc.g[k] * (t0 + t1);
            a1 = a1 + pc.xg[k] * (t1 - t0);
            a2 = a2 + pc.xxg[k] * (t0 + t1);
        }
        float(*rb)[256] = row[r & 1];
        rb[0][tx] = a0;
        rb[1][tx] = a1;
        rb[2][tx] = a2;
        __syncthreads(); // buffer r & 1 is rewritten at row r + 2, after every thread has passed the barrier of row r + 1
        if (writes) {
            float b1 = pc.g[0] * rb[0][tx];
            float b3 = pc.g[0] * rb[1][tx];
            float b5 = pc.g[0] * rb[2][tx];
            float b2 = 0, b4 = 0, b6 = 0;
#pragma unroll
            for (int k = 1; k <= N; ++k) {
                b1 = b1 + (rb[0][tx + k] + rb[0][tx - k]) * pc.g[k];
                b4 = b4 + (rb[0][tx + k] + rb[0][tx - k]) * pc.xxg[k];
                b2 = b2 + (rb[0][tx + k] - rb[0][tx - k]) * pc.xg[k];
                b3 = b3 + (rb[1][tx + k] + rb[1][tx - k]) * pc.g[k];
                b6 = b6 + (rb[1][tx + k] - rb[1][tx - k]) * pc.xg[k];
                b5 = b5 + (rb[2][tx + k] + rb[2][tx - k]) * pc.g[k];
            }
            float *R = Rbase + (long long)y * L.pitch;
            R[0] = b3 * pc.ig11;
            R[ps] = b2 * pc.ig11;
            R[2 * ps] = b1 * pc.ig03 + b5 * pc.ig33;
            R[3 * ps] = b1 * pc.ig03 + b4 * pc.ig33;
            R[4 * ps] = b6 * pc.ig55;
        }
#pragma unroll
        for (int k = 0; k < 2 * N; ++k)
            win[k] = win[k + 1];
        win[2 * N] = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
// per-pair kernels

__device__ __forceinline__ float resize_linear_px_f(const float *src, int sw, int sh, int spitch, int dx, int dy,
                                                    float ifx, float ify) {
    const float sx = (float)dx * ifx;
    const float sy = (float)dy * ify;
    const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
    const int x2 = x1 + 1, y2 = y1 + 1;
    const int x2r = min(x2, sw - 1), y2r = min(y2, sh - 1);
    const int x1r = min(x1, sw - 1), y1r = min(y1, sh - 1);
    float out = 0.0f;
    out = out + src[(long long)y1r * spitch + x1r] * (((float)x2 - sx) * ((float)y2 - sy));
    out = out + src[(long long)y1r * spitch + x2r] * ((sx - (float)x1) * ((float)y2 - sy));
    out = out + src[(long long)y2r * spitch + x1r] * (((float)x2 - sx) * (sy - (float)y1));
    out = out + src[(long long)y2r * spitch + x2r] * ((sx - (float)x1) * (sy - (float)y1));
    return out;
}

__global__ __launch_bounds__(256) void k_farn_init_flow(FarnPairCtx c, int cur_set, int prev_w, int prev_h,
                                                        int prev_pitch, float ifx, float ify, float up, int zero) {
    const DfxBlockXY blk = dfx_block_xy(); // XCD-aware (dfx_device.h): neighbouring rows share one L2
    const int x = blk.x * 64 + (threadIdx.x & 63);
    const int y = blk.y * 4 + (threadIdx.x >> 6);
    if (x >= c.L.w || y >= c.L.h)
        return;
    const int b = blockIdx.z;
    const long long o = (long long)y * c.L.pitch + x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float *dst = farn_plane(c, b, FARN_PL_FX0 + 2 * cur_set + k);
        if (zero) {
            dst[o] = 0.0f;
        } else {
            const float *src = farn_plane(c, b, FARN_PL_FX0 + 2 * (cur_set ^ 1) + k);
            dst[o] = resize_linear_px_f(src, prev_w, prev_h, prev_pitch, x, y, ifx, ify) * up;
        }
    }
}

__constant__ float c_farn_border[6] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f, 1.f};

// 8-byte load with 4-byte alignment: the two horizontal taps of a bilinear sample
typedef float float2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float float2_a8 __attribute__((ext_vector_type(2), aligned(8)));

// B.7 for one pixel.  R0/R1 point at plane 0 of the level inside the two frame slots.
__device__ __forceinline__ void update_matrices_px(const float *R0, const float *R1, int w, int h, int pitch, int x,
                                                   int y, float dx, float dy, float (&M)[5]) {
    const long long ps = (long long)pitch * h;
    const long long o = (long long)y * pitch + x;
    float fx = (float)x + dx;
    float fy = (float)y + dy;
    const int x1 = (int)floorf(fx);
    const int y1 = (int)floorf(fy);
    fx -= (float)x1;
    fy -= (float)y1;
    float r2, r3, r4, r5, r6;
    if (x1 >= 0 && y1 >= 0 && x1 < w - 1 && y1 < h - 1) {
        const float a00 = (1.f - fx) * (1.f - fy);
        const float a01 = fx * (1.f - fy);
        const float a10 = (1.f - fx) * fy;
        const float a11 = fx * fy;
        const long long q = (long long)y1 * pitch + x1;
        float v[5];
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const float *Rp = R1 + p * ps;
            const float2_a4 t0 = *reinterpret_cast<const float2_a4 *>(Rp + q);
            const float2_a4 t1 = *reinterpret_cast<const float2_a4 *>(Rp + q + pitch);
            v[p] = ((a00 * t0[0] + a01 * t0[1]) + a10 * t1[0]) + a11 * t1[1];
        }
        r2 = v[0];
        r3 = v[1];
        r4 = (R0[2 * ps + o] + v[2]) * 0.5f;
        r5 = (R0[3 * ps + o] + v[3]) * 0.5f;
        r6 = (R0[4 * ps + o] + v[4]) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = R0[2 * ps + o];
        r5 = R0[3 * ps + o];
        r6 = R0[4 * ps + o] * 0.5f;
    }
    r2 = (R0[o] - r2) * 0.5f;
    r3 = (R0[ps + o] - r3) * 0.5f;
    r2 = r2 + (r4 * dy + r6 * dx); // r2 += r4*dy + r6*dx
    r3 = r3 + (r6 * dy + r5 * dx);
    float scale = c_farn_border[min(x, 5)] * c_farn_border[min(y, 5)];
    scale = scale * c_farn_border[min(w - x - 1, 5)];
    scale = scale * c_farn_border[min(h - y - 1, 5)];
    r2 *= scale;
    r3 *= scale;
    r4 *= scale;
    r5 *= scale;
    r6 *= scale;
    M[0] = r4 * r4 + r6 * r6;
    M[1] = (r4 + r5) * r6;
    M[2] = r5 * r5 + r6 * r6;
    M[3] = r4 * r2 + r6 * r3;
    M[4] = r6 * r2 + r5 * r3;
}

__global__ __launch_bounds__(256) void k_farn_update_matrices(FarnPairCtx c, int flow_set, int m_set) {
    const DfxBlockXY blk = dfx_block_xy(); // bilinear gathers of R1 around (x + dx, y + dy): neighbours share rows
    const int x = blk.x * 64 + (threadIdx.x & 63);
    const int y = blk.y * 4 + (threadIdx.x >> 6);
    if (x >= c.L.w || y >= c.L.h)
        return;
    const int b = blockIdx.z;
    const PairDesc pd = c.pairs[b];
    const float *R0 = c.frame_R + (long long)pd.frame_a * c.frame_stride + c.L.r_off;
    const float *R1 = c.frame_R + (long long)pd.frame_b * c.frame_stride + c.L.r_off;
    const long long o = (long long)y * c.L.pitch + x;
    const float dx = farn_plane(c, b, FARN_PL_FX0 + 2 * flow_set)[o];
    const float dy = farn_plane(c, b, FARN_PL_FY0 + 2 * flow_set)[o];
    float M[5];
    update_matrices_px(R0, R1, c.L.w, c.L.h, c.L.pitch, x, y, dx, dy, M);
#pragma unroll
    for (int p = 0; p < 5; ++p)
        farn_plane(c, b, (m_set ? FARN_PL_M1 : FARN_PL_M0) + p)[o] = M[p];
}

// One Farneback iteration in one launch: 13x13 box filter of the 5 M planes (B.8: vertical sums, then
// horizontal sums, in upstream's order, replicate border), the 2x2 solve (B.9) and, unless this is
// the last iteration, the next M (B.7) written to the other M set.
// Tile: 64 x 16 output pixels per workgroup, 4 rows per thread; LDS: (64+2h) x (16+2h) input tile and
// 16 x (64+2h) vertical sums, reused for the 5 planes.
__global__ __launch_bounds__(256) void k_farn_iteration(FarnPairCtx c, int flow_set, int m_src, int half,
                                                        float box_inv, int do_matrices) {
    constexpr int TW = 64, TH = 16, HM = FARN_HALF_MAX;
    __shared__ float tile[TH + 2 * HM][TW + 2 * HM];
    __shared__ float vs[TH][TW + 2 * HM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int w = c.L.w, h = c.L.h, pitch = c.L.pitch;
    const int twp = TW + 2 * half, thp = TH + 2 * half;

    float m[5][4];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const float *Mp = farn_plane(c, b, (m_src ? FARN_PL_M1 : FARN_PL_M0) + p);
        for (int ty = wave; ty < thp; ty += 4) {
            const long long ro = (long long)min(max(y0 - half + ty, 0), h - 1) * pitch;
            for (int tx = lane; tx < twp; tx += 64)
                tile[ty][tx] = Mp[ro + min(max(x0 - half + tx, 0), w - 1)];
        }
        __syncthreads();
        for (int r = wave; r < TH; r += 4) {
            for (int tx = lane; tx < twp; tx += 64) {
                float v = tile[r + half][tx];
                for (int j = 1; j <= half; ++j)
                    v = v + (tile[r + half - j][tx] + tile[r + half + j][tx]);
                vs[r][tx] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 4 + i;
            float res = vs[r][lane + half];
            for (int k = 1; k <= half; ++k)
                res = res + (vs[r][lane + half - k] + vs[r][lane + half + k]);
            m[p][i] = res * box_inv;
        }
        __syncthreads();
    }

    const int x = x0 + lane;
    if (x >= w)
        return;
    const PairDesc pd = c.pairs[b];
    const float *R0 = c.frame_R + (long long)pd.frame_a * c.frame_stride + c.L.r_off;
    const float *R1 = c.frame_R + (long long)pd.frame_b * c.frame_stride + c.L.r_off;
    float *FX = farn_plane(c, b, FARN_PL_FX0 + 2 * flow_set), *FY = farn_plane(c, b, FARN_PL_FY0 + 2 * flow_set);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + wave * 4 + i;
        if (y >= h)
            break;
        const long long o = (long long)y * pitch + x;
        const float g11 = m[0][i], g12 = m[1][i], g22 = m[2][i], h1 = m[3][i], h2 = m[4][i];
        const float detInv = 1.f / ((g11 * g22 - g12 * g12) + 1e-3f);
        const float fx = (g11 * h2 - g12 * h1) * detInv;
        const float fy = (g22 * h1 - g12 * h2) * detInv;
        FX[o] = fx;
        FY[o] = fy;
        if (do_matrices) {
            float M[5];
            update_matrices_px(R0, R1, w, h, pitch, x, y, fx, fy, M);
#pragma unroll
            for (int p = 0; p < 5; ++p)
                farn_plane(c, b, (m_src ? FARN_PL_M0 : FARN_PL_M1) + p)[o] = M[p];
        }
    }
}

// The same iteration for a compile-time box half-width, restructured around LDS traffic and latency:
//   * 64 x 32 output pixels per workgroup: the halo re-read of M drops from 2.08x to 1.63x;
//   * the halo tile of plane p+1 is fetched into registers while plane p is being summed;
//   * vertical sums: one thread owns a 10- or 12-row strip of one tile column and keeps its inputs in registers
//     (about two LDS reads per sum instead of 2*HALF + 1), one pass, one strip per wave: no bank conflicts (round 3;
//     measured equal to round 2's two-pass form within the run-to-run noise — the kernel is bound by its HBM
//     traffic, 62 B per pixel at ~4.5 TB/s, not by LDS: profiles/round3/experiments/farn_iteration_kernel_ab.txt —
//     kept because it needs 110 instead of 126 VGPRs and is the simpler code);
//   * the halo tile is fetched as 8-byte column pairs wherever its columns need no clamping;
//   * horizontal sums: one thread owns an 8-column strip of one row, read with 16-byte LDS loads (row pitch
//     = 4 mod 32 words keeps the lanes of a wave, which walk down the rows, on distinct banks); the 8 results
//     go back through a small LDS array so that the solve / updateMatrices phase sees lane = x again.
// Every sum is still  centre + (left_1 + right_1) + (left_2 + right_2) ...  in upstream's order (B.8).
template <int HALF>
__global__ __launch_bounds__(256) void k_farn_iteration_t(FarnPairCtx c, int flow_set, int m_src, float box_inv,
                                                          int do_matrices) {
    constexpr int TW = 64, TH = 32, IW = TW + 2 * HALF, IH = TH + 2 * HALF;
    static_assert(HALF == 6 && TH == 32, "the strip / bank layout of the vertical pass is worked out for a 76 x 44 halo tile");
    constexpr int VP = ((IW + 27) / 32) * 32 + 4; // pitch of the vertical-sum rows: >= IW, = 4 mod 32, 16-B rows
    constexpr int HP = TW + 4;                    // pitch of the result rows
    __shared__ __attribute__((aligned(16))) float tile[IH][IW];
    __shared__ __attribute__((aligned(16))) float vs[TH][VP];
    __shared__ __attribute__((aligned(16))) float hb[TH][HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const DfxBlockXY blk = dfx_block_xy(); // the 76 x 44 halo tiles of neighbours overlap (M is read 1.63 x): one L2
    const int x0 = blk.x * TW, y0 = blk.y * TH;
    const int w = c.L.w, h = c.L.h, pitch = c.L.pitch;
    const float *Mbase = farn_plane(c, b, m_src ? FARN_PL_M1 : FARN_PL_M0);

    // The halo tile is fetched as pairs of columns.  x0 - HALF is even and the pitch a multiple of 64, so wherever the
    // 76 halo columns need no clamping (every tile but the first / last of a row) a pair is ONE 8-byte load.  Tiles that
    // touch the left / right border clamp per element and recompute their offsets for every plane.
    constexpr int NE2 = IW / 2 * IH;        // column pairs of the halo tile
    constexpr int NLD2 = (NE2 + 255) / 256; // pairs per thread
    const bool xin = x0 >= HALF && x0 + TW + HALF <= w;
    int off2[NLD2];
#pragma unroll
    for (int k = 0; k < NLD2; ++k) {
        const int e = min(tid + k * 256, NE2 - 1);
        const int ty = e / (IW / 2), tx = 2 * (e - ty * (IW / 2));
        off2[k] = min(max(y0 - HALF + ty, 0), h - 1) * pitch + (x0 - HALF + tx); // rows clamp, columns do not (xin)
    }
    float pre[2 * NLD2];
    auto fetch = [&](const float *Mp) {
        if (xin) {
#pragma unroll
            for (int k = 0; k < NLD2; ++k) {
                const float2_a8 t = *reinterpret_cast<const float2_a8 *>(Mp + off2[k]);
                pre[2 * k] = t[0];
                pre[2 * k + 1] = t[1];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD2; ++k) {
                const int e = min(tid + k * 256, NE2 - 1);
                const int ty = e / (IW / 2), tx = 2 * (e - ty * (IW / 2));
                const int ro = min(max(y0 - HALF + ty, 0), h - 1) * pitch;
                pre[2 * k] = Mp[ro + min(max(x0 - HALF + tx, 0), w - 1)];
                pre[2 * k + 1] = Mp[ro + min(max(x0 - HALF + tx + 1, 0), w - 1)];
            }
        }
    };
    auto stash = [&]() { // pair e of the tile is floats 2e, 2e + 1 of its row-major image (IW is even)
#pragma unroll
        for (int k = 0; k < NLD2; ++k)
            if (tid + k * 256 < NE2) {
                float2_a8 t;
                t[0] = pre[2 * k];
                t[1] = pre[2 * k + 1];
                *reinterpret_cast<float2_a8 *>(&tile[0][0] + 2 * (tid + k * 256)) = t;
            }
    };
    fetch(Mbase);
    stash();
    __syncthreads();

    // Vertical sums of one column over a strip of NR rows starting at tile row r0: NR + 2 * HALF inputs in registers
    // (about two LDS reads per sum instead of 13), upstream's order (centre + (up_1 + down_1) + (up_2 + down_2) ...).
    auto vstrip = [&](auto nr, int r0, int col) {
        constexpr int NR = decltype(nr)::value;
        float v[NR + 2 * HALF];
#pragma unroll
        for (int j = 0; j < NR + 2 * HALF; ++j)
            v[j] = tile[r0 + j][col];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float a = v[i + HALF];
#pragma unroll
            for (int j = 1; j <= HALF; ++j)
                a = a + (v[i + HALF - j] + v[i + HALF + j]);
            vs[r0 + i][col] = a;
        }
    };
    // ONE pass: waves 0, 1, 2 take columns 0..63 of three strips of 12 / 10 / 10 rows (one strip per wave: consecutive
    // lanes, consecutive banks), wave 3 the 12 halo columns 64..75 of all three strips — as 12-row items starting at rows
    // 0 / 10 / 20 so that its lanes run one code path (rows 10, 11, 20, 21 are written twice with the same values),
    // placed so that items of one half-wave are >= 12 banks apart: rows 0 and 20 (20 * 76 = 16 mod 32) in lanes 0-11 /
    // 12-23, row 10 (10 * 76 = 24 mod 32) in lanes 32-43.  (Round 2's second pass kept 48 lanes busy, and every wave but
    // the first straddled two strips whose tile rows are 8 * 76 = 0 (mod 32) words apart: two-way bank conflicts.)
    const int v3_row0 = lane < 12 ? 0 : lane < 24 ? 20 : (lane >= 32 && lane < 44) ? 10 : -1;
    const int v3_col = TW + (lane < 12 ? lane : lane < 24 ? lane - 12 : lane - 32);

    float m[5][8];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        if (p + 1 < 5) // in flight during the vertical pass
            fetch(Mbase + (long long)(p + 1) * c.plane_stride);
        if (wave == 0)
            vstrip(std::integral_constant<int, 12>{}, 0, lane);
        else if (wave < 3)
            vstrip(std::integral_constant<int, 10>{}, 2 + 10 * wave, lane);
        else if (v3_row0 >= 0)
            vstrip(std::integral_constant<int, 12>{}, v3_row0, v3_col);
        __syncthreads(); // A: vs complete, tile free
        if (p + 1 < 5)
            stash();
        { // horizontal sums: row = tid & 31, columns (tid >> 5) * 8 .. + 8
            const int row = tid & 31, cs = (tid >> 5) * 8;
            float v[8 + 2 * HALF + 3];
            const float4 *src = reinterpret_cast<const float4 *>(&vs[row][cs]);
#pragma unroll
            for (int q = 0; q < (8 + 2 * HALF + 3) / 4; ++q) {
                const float4 t = src[q];
                v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
            }
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a = v[i + HALF];
#pragma unroll
                for (int k = 1; k <= HALF; ++k)
                    a = a + (v[i + HALF - k] + v[i + HALF + k]);
                r[i] = a * box_inv;
            }
            float4 *dst = reinterpret_cast<float4 *>(&hb[row][cs]);
            dst[0] = make_float4(r[0], r[1], r[2], r[3]);
            dst[1] = make_float4(r[4], r[5], r[6], r[7]);
        }
        __syncthreads(); // B: hb complete, vs free, next tile visible
#pragma unroll
        for (int i = 0; i < 8; ++i)
            m[p][i] = hb[wave * 8 + i][lane];
    }

    const int x = x0 + lane;
    if (x >= w)
        return;
    const PairDesc pd = c.pairs[b];
    const float *R0 = c.frame_R + (long long)pd.frame_a * c.frame_stride + c.L.r_off;
    const float *R1 = c.frame_R + (long long)pd.frame_b * c.frame_stride + c.L.r_off;
    float *FX = farn_plane(c, b, FARN_PL_FX0 + 2 * flow_set), *FY = farn_plane(c, b, FARN_PL_FY0 + 2 * flow_set);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int y = y0 + wave * 8 + i;
        if (y >= h)
            break;
        const long long o = (long long)y * pitch + x;
        const float g11 = m[0][i], g12 = m[1][i], g22 = m[2][i], h1 = m[3][i], h2 = m[4][i];
        const float detInv = 1.f / ((g11 * g22 - g12 * g12) + 1e-3f);
        const float fx = (g11 * h2 - g12 * h1) * detInv;
        const float fy = (g22 * h1 - g12 * h2) * detInv;
        FX[o] = fx;
        FY[o] = fy;
        if (do_matrices) {
            float M[5];
            update_matrices_px(R0, R1, w, h, pitch, x, y, fx, fy, M);
#pragma unroll
            for (int p = 0; p < 5; ++p)
                farn_plane(c, b, (m_src ? FARN_PL_M0 : FARN_PL_M1) + p)[o] = M[p];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the iteration with M never in HBM (VERDICT r3 "Next round" item 4) — k_farn_iter_stream below, and the
// device functions it is made of.
//
// k_farn_iteration_t reads the five M planes of the previous launch with a 6-pixel halo and writes the next five: 40 of
// SURVEY.md section 8d's 136 algorithmic B/px per iteration and, measured at the bench's batch of 129 pairs, 90 B/px of
// real HBM traffic at ~5 TB/s — it is HBM-bound.  But B.7's updateMatrices is a POINTWISE function of (flow, R0, gathered
// R1) at a pixel, so M can be recomputed where the box filter needs it from the previous launch's FLOW (8 B/px instead
// of 20, and nothing written back).  The flow then ping-pongs between its two plane sets (a workgroup reads its
// neighbours' previous flow while they write the next one) and the first k_farn_update_matrices launch of a level
// disappears.
//
// The price is VALU and gather work on the halo, so the arithmetic is written for v_pk_*_f32 (two IEEE operations per
// lane and issue slot, each half rounded on its own — bit-identical to the scalar form; packed f32 issues at the scalar
// rate on gfx950): LDS holds (M0, M2) and (M3, M4) as float2 planes and M1 alone, which makes every box-filter addition a
// packed addition of two planes (M1: of two columns in the vertical pass), and updateMatrices itself pairs its products.
// Every sum keeps upstream's order (centre + (l1 + r1) + (l2 + r2) ...).  Same operations on the same inputs in the same
// order => bit-identical flows: tests/test_farneback_gpu.py holds this kernel, the M-in-HBM kernel
// (dfx_params.variant & DFX_VAR_FARN_M_IN_HBM), the simple kernels (impl = 1) and the oracle to each other.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Loads at a uniform base + a 32-bit BYTE offset: the form the compiler turns into `global_load ... v_off, s[base]`
// (one VGPR of address per load instead of a 64-bit add per load; an element offset would have to be widened before the
// * 4 and falls back to 64-bit VALU address arithmetic).
__device__ __forceinline__ float farn_ld1(const float *base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ f2 farn_ld2u(const float *base, unsigned byte_off) { // 4-byte aligned pair
    const float2_a4 t = *reinterpret_cast<const float2_a4 *>(reinterpret_cast<const char *>(base) + byte_off);
    f2 r;
    r.x = t[0], r.y = t[1];
    return r;
}
__device__ __forceinline__ f2 farn_ld2(const float *base, unsigned byte_off) { // 8-byte aligned pair
    const float2_a8 t = *reinterpret_cast<const float2_a8 *>(reinterpret_cast<const char *>(base) + byte_off);
    f2 r;
    r.x = t[0], r.y = t[1];
    return r;
}
__device__ __forceinline__ f2 farn_f2(float a, float b) {
    f2 r;
    r.x = a, r.y = b;
    return r;
}

// c_farn_border[min(d, 5)] without the table (a per-lane constant-memory load each): d >= 0
__device__ __forceinline__ float farn_border_w(int d) { return d < 2 ? 0.14f : (d < 5 ? 0.4472f : 1.f); }

// B.7 for the halo tile, update_matrices_px operation for operation, split in three so that the two pixels of a work item
// can share their R1 loads: the sample position, the bilinear sample of one plane, and everything after the samples.
struct FarnGeo {
    int x1, y1;   // top-left tap
    f2 w0, w1;    // (a00, a01), (a10, a11)
    bool valid;   // all four taps inside the image (B.7's test)
};
__device__ __forceinline__ FarnGeo farn_geo(int x, int y, float dx, float dy, int w, int h) {
    FarnGeo g;
    float fx = (float)x + dx;
    float fy = (float)y + dy;
    g.x1 = (int)floorf(fx);
    g.y1 = (int)floorf(fy);
    fx -= (float)g.x1;
    fy -= (float)g.y1;
    g.valid = (unsigned)g.x1 < (unsigned)(w - 1) && (unsigned)g.y1 < (unsigned)(h - 1); // x1, y1 >= 0, x1 < w-1, y1 < h-1
    const f2 wx = farn_f2(1.f - fx, fx);
    g.w0 = wx * (1.f - fy);
    g.w1 = wx * fy;
    return g;
}
// ((a00 * t00 + a01 * t01) + a10 * t10) + a11 * t11 with the four products as two packed multiplications
__device__ __forceinline__ float farn_bilinear(const FarnGeo &g, f2 row0, f2 row1) {
    const f2 t0 = g.w0 * row0, t1 = g.w1 * row1;
    return ((t0.x + t0.y) + t1.x) + t1.y;
}
// Everything after the samples (v[] is ignored where !valid).  Out: m02 = (M0, M2), m34 = (M3, M4), m1 = M1.
__device__ __forceinline__ void farn_um_finish(bool valid, const float (&v)[5], const float (&r0)[5], float dx, float dy,
                                               int x, int y, int w, int h, bool unit_scale, f2 &m02, f2 &m34, float &m1) {
    const float r2s = valid ? v[0] : 0.f, r3s = valid ? v[1] : 0.f;
    const f2 r45v = (farn_f2(r0[2], r0[3]) + farn_f2(v[2], v[3])) * 0.5f;
    f2 r45 = farn_f2(valid ? r45v.x : r0[2], valid ? r45v.y : r0[3]);
    float r6 = valid ? (r0[4] + v[4]) * 0.25f : r0[4] * 0.5f;
    f2 r23 = (farn_f2(r0[0], r0[1]) - farn_f2(r2s, r3s)) * 0.5f;
    r23 = r23 + (farn_f2(r45.x, r6) * dy + farn_f2(r6, r45.y) * dx); // r2 += r4*dy + r6*dx ; r3 += r6*dy + r5*dx
    if (!unit_scale) {
        float scale = farn_border_w(min(x, 5)) * farn_border_w(min(y, 5));
        scale = scale * farn_border_w(min(w - x - 1, 5));
        scale = scale * farn_border_w(min(h - y - 1, 5));
        r23 = r23 * scale;
        r45 = r45 * scale;
        r6 *= scale;
    }
    const float r66 = r6 * r6;
    m02 = r45 * r45 + farn_f2(r66, r66);                           // r4*r4 + r6*r6 ; r5*r5 + r6*r6
    m1 = (r45.x + r45.y) * r6;                                     // (r4 + r5) * r6
    m34 = farn_f2(r45.x, r6) * r23.x + farn_f2(r6, r45.y) * r23.y; // r4*r2 + r6*r3 ; r6*r2 + r5*r3
}

typedef float float4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f4 farn_ld4u(const float *base, unsigned byte_off) { // 4-byte aligned quad
    const float4_a4 t = *reinterpret_cast<const float4_a4 *>(reinterpret_cast<const char *>(base) + byte_off);
    f4 r;
    r.x = t[0], r.y = t[1], r.z = t[2], r.w = t[3];
    return r;
}

// The five bilinear samples of R1 for the TWO horizontally adjacent pixels of a work item.  The kernel is bound by the
// texture-address unit (TA busy 76 % of the time with one 8-byte gather per tap row: profiles/round4/farn_iter/), and the
// second pixel's taps are, wherever the flow is smooth, the first pixel's shifted by one column: then ONE 16-byte load per
// row and plane (columns x1 .. x1 + 3) serves both pixels — half the gather instructions.  `joint` = both pixels valid, same
// tap row, second tap column 0 .. 2 right of the first, window inside the row; other lanes take the per-pixel gathers.
// Same values into the same arithmetic either way.
__device__ __forceinline__ void farn_sample_pair(const float *__restrict__ R1, unsigned ps, int w, int pitch,
                                                 const FarnGeo &ga, const FarnGeo &gb, float (&va)[5], float (&vb)[5]) {
    const int off = gb.x1 - ga.x1;
    const bool joint = ga.valid && gb.valid && gb.y1 == ga.y1 && (unsigned)off <= 2u && ga.x1 + 3 <= w - 1;
    const unsigned qa = (unsigned)(ga.y1 * pitch + ga.x1) * 4u, qa1 = qa + (unsigned)pitch * 4u;
    if (__all(joint && off == 1)) { // the whole wave: static taps
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const float *Rp = R1 + p * ps;
            const f4 t0 = farn_ld4u(Rp, qa), t1 = farn_ld4u(Rp, qa1);
            va[p] = farn_bilinear(ga, farn_f2(t0.x, t0.y), farn_f2(t1.x, t1.y));
            vb[p] = farn_bilinear(gb, farn_f2(t0.y, t0.z), farn_f2(t1.y, t1.z));
        }
        return;
    }
    if (joint) {
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const float *Rp = R1 + p * ps;
            const f4 t0 = farn_ld4u(Rp, qa), t1 = farn_ld4u(Rp, qa1);
            va[p] = farn_bilinear(ga, farn_f2(t0.x, t0.y), farn_f2(t1.x, t1.y));
            const f2 b0 = off == 0 ? farn_f2(t0.x, t0.y) : off == 1 ? farn_f2(t0.y, t0.z) : farn_f2(t0.z, t0.w);
            const f2 b1 = off == 0 ? farn_f2(t1.x, t1.y) : off == 1 ? farn_f2(t1.y, t1.z) : farn_f2(t1.z, t1.w);
            vb[p] = farn_bilinear(gb, b0, b1);
        }
    } else {
        if (ga.valid) {
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                const float *Rp = R1 + p * ps;
                va[p] = farn_bilinear(ga, farn_ld2u(Rp, qa), farn_ld2u(Rp, qa1));
            }
        }
        if (gb.valid) {
            const unsigned qb = (unsigned)(gb.y1 * pitch + gb.x1) * 4u, qb1 = qb + (unsigned)pitch * 4u;
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                const float *Rp = R1 + p * ps;
                vb[p] = farn_bilinear(gb, farn_ld2u(Rp, qb), farn_ld2u(Rp, qb1));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The iteration as a STREAM down a column strip.  A workgroup (4 waves) owns 64 output columns of a segment of rows and
// walks down it 6 rows at a time with a ring of 18 M rows in LDS (27 KB):
//     step:  updateMatrices for the 6 rows entering the window (76 columns: halo factor 1.19)   -> ring
//            vertical 13-sums of the 6 output rows, one column per lane, written over the 6 rows leaving the window
//            horizontal 13-sums + the 2x2 solve for 6 x 64 pixels                               -> flow_out
// (The first form of this round recomputed M on 64 x 32 tiles with a 6-pixel halo — 1.63 x the pixels, 68 KB of LDS, two
// workgroups per CU whose gather and summing phases did not overlap: 711 + 323 = 1027 us per launch,
// profiles/round4/farn_iter/.  The stream reads every input row once per strip and measures 965 us.)
#ifndef FARN_STREAM_WPS
#define FARN_STREAM_WPS 4
#endif
#ifndef FARN_STREAM_MASK
#define FARN_STREAM_MASK 3 // measurement builds: bit 0 = updateMatrices, bit 1 = sums + solve
#endif
// INIT: the first iteration of a level.  Its input flow is not a plane set of this level but the previous (coarser)
// level's final flow, up-sampled on the fly — resize_linear_px_f(...) * (1 / pyrScale), k_farn_init_flow's very
// expression — or zero at the coarsest level: the init launch and its 16 B/px of traffic per level disappear.  flow_in
// then names the set that holds the coarser level's flow (with that level's geometry, FarnInit).
struct FarnInit {
    int zero;                 // coarsest level: flow = 0
    int prev_w, prev_h, prev_pitch;
    float ifx, ify, up;
};
template <int HALF, bool INIT>
__global__ __launch_bounds__(256, FARN_STREAM_WPS) void k_farn_iter_stream(FarnPairCtx c, int flow_in, int flow_out, float box_inv,
                                                             int seg_rows, float *merged, long long merged_stride,
                                                             FarnInit init) {
    constexpr int TW = 64, IW = TW + 2 * HALF, RB = 6, RING = RB + 2 * HALF, NP = IW / 2; // 38 column pairs per row
    static_assert(HALF == 6 && RING == 18 && RB * NP <= 256 && RB * (TW / 2) <= 256, "work split worked out for 6-row steps");
    __shared__ __attribute__((aligned(16))) f2 A[RING][IW];    // (M0, M2)
    __shared__ __attribute__((aligned(16))) f2 B[RING][IW];    // (M3, M4)
    __shared__ __attribute__((aligned(16))) float C[RING][IW]; // M1
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const DfxBlockXY blk = dfx_block_xy();
    const int x0 = blk.x * TW;
    const int w = c.L.w, h = c.L.h, pitch = c.L.pitch;
    const int ya = blk.y * seg_rows, yb = min(ya + seg_rows, h);
    if (ya >= h)
        return;
    const unsigned ps = (unsigned)pitch * (unsigned)h;
    const PairDesc pd = c.pairs[b];
    const float *R0 = c.frame_R + (long long)pd.frame_a * c.frame_stride + c.L.r_off;
    const float *R1 = c.frame_R + (long long)pd.frame_b * c.frame_stride + c.L.r_off;
    const float *FXi = farn_plane(c, b, FARN_PL_FX0 + 2 * flow_in), *FYi = farn_plane(c, b, FARN_PL_FY0 + 2 * flow_in);
    float *FXo = farn_plane(c, b, FARN_PL_FX0 + 2 * flow_out), *FYo = farn_plane(c, b, FARN_PL_FY0 + 2 * flow_out);

    const bool xin = x0 >= HALF && x0 + TW + HALF <= w;
    const bool unit_cols = x0 - HALF >= 5 && x0 + TW + HALF - 1 <= w - 6; // border attenuation 1 in x for every column
    // updateMatrices work item of this thread: row k of the 6 entering rows, column pair pr
    const bool has_item = tid < RB * NP;
    const int k1 = has_item ? tid / NP : 0, tx = 2 * ((has_item ? tid : 0) - k1 * NP);
    const int gxa = x0 - HALF + tx;
    int gx[2];
    gx[0] = xin ? gxa : min(max(gxa, 0), w - 1);
    gx[1] = xin ? gxa + 1 : min(max(gxa + 1, 0), w - 1);
    auto load_flow = [&](int gy, f2 &dx, f2 &dy) {
        if (INIT) {
            if (init.zero) {
                dx = dy = farn_f2(0.f, 0.f);
            } else {
                dx = farn_f2(resize_linear_px_f(FXi, init.prev_w, init.prev_h, init.prev_pitch, gx[0], gy, init.ifx, init.ify) * init.up,
                             resize_linear_px_f(FXi, init.prev_w, init.prev_h, init.prev_pitch, gx[1], gy, init.ifx, init.ify) * init.up);
                dy = farn_f2(resize_linear_px_f(FYi, init.prev_w, init.prev_h, init.prev_pitch, gx[0], gy, init.ifx, init.ify) * init.up,
                             resize_linear_px_f(FYi, init.prev_w, init.prev_h, init.prev_pitch, gx[1], gy, init.ifx, init.ify) * init.up);
            }
        } else if (xin) {
            const unsigned o = (unsigned)(gy * pitch + gx[0]) * 4u;
            dx = farn_ld2(FXi, o);
            dy = farn_ld2(FYi, o);
        } else {
            const unsigned o0 = (unsigned)(gy * pitch + gx[0]) * 4u, o1 = (unsigned)(gy * pitch + gx[1]) * 4u;
            dx = farn_f2(farn_ld1(FXi, o0), farn_ld1(FXi, o1));
            dy = farn_f2(farn_ld1(FYi, o0), farn_ld1(FYi, o1));
        }
    };
    // The R0 values of the item's row (like its flow) are fetched one step AHEAD of their use and held in registers across
    // the sums of the step before: 10 registers that keep loads in flight while the workgroup sums.  The R1 windows are
    // loaded when they are used — holding them too (40 registers) costs the fourth workgroup per CU and measures 3 % slower
    // (166 registers, 3 workgroups: 988-995 us per launch; this form, 126 registers, 4 workgroups: 962-967 us;
    // profiles/round4/farn_iter/stream_prefetch_rates.txt).
    float pr0[2][5];
    auto issue_row = [&](int gy) {
        if (xin) {
            const unsigned o = (unsigned)(gy * pitch + gx[0]) * 4u;
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                const f2 t = farn_ld2(R0 + p * ps, o);
                pr0[0][p] = t.x, pr0[1][p] = t.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned o = (unsigned)(gy * pitch + gx[j]) * 4u;
#pragma unroll
                for (int p = 0; p < 5; ++p)
                    pr0[j][p] = farn_ld1(R0 + p * ps, o);
            }
        }
    };
    // M of the item's two pixels at image row gy (already clamped) from the prefetched R0, into ring slot `slot`
    auto matrices_row = [&](int gy, f2 dx, f2 dy, int slot) {
        const bool unit_scale = unit_cols && gy >= 5 && gy <= h - 6;
        const FarnGeo ga = farn_geo(gx[0], gy, dx.x, dy.x, w, h);
        const FarnGeo gb = farn_geo(gx[1], gy, dx.y, dy.y, w, h);
        float va[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, vb[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        farn_sample_pair(R1, ps, w, pitch, ga, gb, va, vb);
        f2 m02[2], m34[2];
        float m1[2];
        farn_um_finish(ga.valid, va, pr0[0], dx.x, dy.x, gx[0], gy, w, h, unit_scale, m02[0], m34[0], m1[0]);
        farn_um_finish(gb.valid, vb, pr0[1], dx.y, dy.y, gx[1], gy, w, h, unit_scale, m02[1], m34[1], m1[1]);
        f4 a, bb;
        a.x = m02[0].x, a.y = m02[0].y, a.z = m02[1].x, a.w = m02[1].y;
        bb.x = m34[0].x, bb.y = m34[0].y, bb.z = m34[1].x, bb.w = m34[1].y;
        *reinterpret_cast<f4 *>(&A[slot][tx]) = a;
        *reinterpret_cast<f4 *>(&B[slot][tx]) = bb;
        *reinterpret_cast<f2 *>(&C[slot][tx]) = farn_f2(m1[0], m1[1]);
    };
    // ring row o (counted from image row ya - HALF) lives in slot o % RING and holds M of image row clamp(ya - HALF + o)
    auto image_row = [&](int o) { return min(max(ya - HALF + o, 0), h - 1); };

    // vertical-sum work item of this thread: one float2 column of A or B (96 lanes each so that a 32-lane LDS read group
    // stays inside one of them), or one column PAIR of the single plane C
    f2 *vcol = nullptr;
    int vstride = 0;
    if (tid < 2 * 96) {
        const int g = tid / 96, col = tid - g * 96;
        if (col < IW) {
            vcol = (g ? &B[0][0] : &A[0][0]) + col;
            vstride = IW;
        }
    } else if (tid - 2 * 96 < NP) {
        vcol = reinterpret_cast<f2 *>(&C[0][0]) + (tid - 2 * 96);
        vstride = NP;
    }
    // horizontal-sum work item: output row hi of the step, pixels hx, hx + 1 of the strip
    const bool has_out = tid < RB * (TW / 2);
    const int hi = tid >> 5, hx = 2 * (tid & 31);

    // the window's first 12 rows, then the loads of step 0's rows and the flows of step 1's
    f2 cdx = farn_f2(0.f, 0.f), cdy = cdx; // flow of the row whose loads are in flight
    f2 ndx = cdx, ndy = cdx;               // flow of the row after that
    if (has_item) {
        f2 dx0, dy0, dx1, dy1;
        load_flow(image_row(k1), dx0, dy0);
        load_flow(image_row(RB + k1), dx1, dy1);
        load_flow(image_row(2 * RB + k1), cdx, cdy);
        issue_row(image_row(k1));
        matrices_row(image_row(k1), dx0, dy0, k1);
        issue_row(image_row(RB + k1));
        matrices_row(image_row(RB + k1), dx1, dy1, RB + k1);
        issue_row(image_row(2 * RB + k1));
        load_flow(image_row(3 * RB + k1), ndx, ndy);
    }
    const int n_steps = (yb - ya + RB - 1) / RB;
    int b0 = 0; // slot of the oldest row of the window = (RB * s) % RING
    for (int s = 0; s < n_steps; ++s) {
        // rows 12 + 6 s .. 17 + 6 s enter the window: their loads were issued a step ago
        if (has_item) {
            int slot = b0 + 2 * RB + k1;
            slot -= slot >= RING ? RING : 0;
#if FARN_STREAM_MASK & 1
            matrices_row(image_row(2 * RB + RB * s + k1), cdx, cdy, slot);
            if (s + 1 < n_steps) { // the next step's loads fly while this step sums; the flows of the step after it, too
                cdx = ndx, cdy = ndy;
                issue_row(image_row(2 * RB + RB * (s + 1) + k1));
                if (s + 2 < n_steps)
                    load_flow(image_row(2 * RB + RB * (s + 2) + k1), ndx, ndy);
            }
#else
            A[slot][tx] = cdx, B[slot][tx] = cdy, C[slot][tx] = cdx.x;
#endif
        }
        __syncthreads();
#if !(FARN_STREAM_MASK & 2)
        if (A[b0][tid & 63].x == 123456.789f)
            FXo[tid] = B[b0][0].x;
        b0 += RB;
        b0 -= b0 >= RING ? RING : 0;
        continue;
#endif
        // vertical sums of the 6 output rows (window rows j = 0 .. 17 -> slots (b0 + j) % 18), this lane's column only:
        // the sums go where the 6 oldest rows were — nobody else touches this column in this phase
        if (vcol) {
            f2 v[RING];
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                int slot = b0 + j;
                slot -= slot >= RING ? RING : 0;
                v[j] = vcol[slot * vstride];
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                f2 acc = v[i + HALF];
#pragma unroll
                for (int j = 1; j <= HALF; ++j)
                    acc = acc + (v[i + HALF - j] + v[i + HALF + j]);
                int slot = b0 + i;
                slot -= slot >= RING ? RING : 0;
                vcol[slot * vstride] = acc;
            }
        }
        __syncthreads();
        // horizontal sums + solve for two pixels of one output row
        const int y = ya + RB * s + hi, x = x0 + hx;
        if (has_out && y < yb && x < w) {
            int slot = b0 + hi;
            slot -= slot >= RING ? RING : 0;
            f2 s02[2], s34[2];
            float s1[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f2 v[2 + 2 * HALF];
                const f4 *src = reinterpret_cast<const f4 *>(q ? &B[slot][hx] : &A[slot][hx]);
#pragma unroll
                for (int t = 0; t < (2 + 2 * HALF) / 2; ++t) {
                    const f4 u = src[t];
                    v[2 * t] = farn_f2(u.x, u.y), v[2 * t + 1] = farn_f2(u.z, u.w);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f2 acc = v[i + HALF];
#pragma unroll
                    for (int kk = 1; kk <= HALF; ++kk)
                        acc = acc + (v[i + HALF - kk] + v[i + HALF + kk]);
                    (q ? s34 : s02)[i] = acc * box_inv;
                }
            }
            {
                float v[2 + 2 * HALF];
                const f2 *src = reinterpret_cast<const f2 *>(&C[slot][hx]);
#pragma unroll
                for (int t = 0; t < (2 + 2 * HALF) / 2; ++t) {
                    const f2 u = src[t];
                    v[2 * t] = u.x, v[2 * t + 1] = u.y;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float acc = v[i + HALF];
#pragma unroll
                    for (int kk = 1; kk <= HALF; ++kk)
                        acc = acc + (v[i + HALF - kk] + v[i + HALF + kk]);
                    s1[i] = acc * box_inv;
                }
            }
            float fxo[2], fyo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float g11 = s02[i].x, g12 = s1[i], g22 = s02[i].y, h1 = s34[i].x, h2 = s34[i].y;
                const float detInv = 1.f / ((g11 * g22 - g12 * g12) + 1e-3f);
                fxo[i] = (g11 * h2 - g12 * h1) * detInv;
                fyo[i] = (g22 * h1 - g12 * h2) * detInv;
            }
            const unsigned o = (unsigned)(y * pitch + x);
            if (merged) {
                // the last iteration of level 0 writes the caller's interleaved (u, v) rows itself (k_farn_merge's job:
                // one launch and 16 B/px of traffic less per pair); w may be odd, so an 8-byte store per pixel
                float2 *dst = reinterpret_cast<float2 *>(merged + (long long)b * merged_stride) + ((long long)y * w + x);
                dst[0] = make_float2(fxo[0], fyo[0]);
                if (x + 1 < w)
                    dst[1] = make_float2(fxo[1], fyo[1]);
            } else if (x + 1 < w) { // x is even and the pitch a multiple of 64: 8-byte stores
                *reinterpret_cast<float2 *>(FXo + o) = make_float2(fxo[0], fxo[1]);
                *reinterpret_cast<float2 *>(FYo + o) = make_float2(fyo[0], fyo[1]);
            } else {
                FXo[o] = fxo[0];
                FYo[o] = fyo[0];
            }
        }
        __syncthreads(); // the 6 slots just read take the next step's new rows
        b0 += RB;
        b0 -= b0 >= RING ? RING : 0;
    }
}

__global__ __launch_bounds__(256) void k_farn_merge(FarnPairCtx c, int flow_set, float *out, long long out_stride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= c.L.w || y >= c.L.h)
        return;
    const int b = blockIdx.z;
    const long long o = (long long)y * c.L.pitch + x;
    float2 v;
    v.x = farn_plane(c, b, FARN_PL_FX0 + 2 * flow_set)[o];
    v.y = farn_plane(c, b, FARN_PL_FY0 + 2 * flow_set)[o];
    reinterpret_cast<float2 *>(out + (long long)b * out_stride)[(long long)y * c.L.w + x] = v;
}

// ------------------------------------------------------------------------------------------------
// launchers

void farn_launch_u8_to_f32(hipStream_t s, const unsigned char *src, long long src_frame_stride, long long src_pitch,
                           int n_frames, float *dst, long long dst_frame_stride, int w, int h, int pitch) {
    hipLaunchKernelGGL(k_farn_u8_to_f32, grid64x4(w, h, n_frames), dim3(256), 0, s, src, src_frame_stride, src_pitch,
                       dst, dst_frame_stride, w, h, pitch);
}

// Zero-weight taps of the pyramid resize are not evaluated (k_farn_blur_v / k_farn_blur_h_resize) and the polynomial
// expansion walks 16 rows per workgroup: the defaults of the engine's two frame-preparation switches.
#ifndef FARN_POLYROWS_DEFAULT
#define FARN_POLYROWS_DEFAULT 16
#endif
#ifndef FARN_SKIP0_DEFAULT
#define FARN_SKIP0_DEFAULT 1
#endif
int farn_skip_zero_weights_default() { return FARN_SKIP0_DEFAULT; }
int farn_polyexp_rows_default() { return FARN_POLYROWS_DEFAULT; }

void farn_launch_blur_v(hipStream_t s, const float *frames, long long frame_stride, int n_frames, int W, int H,
                        int pitch0, int dst_h, float ify, const float *ker_half, int half, float *tmpv,
                        long long tmpv_frame_stride, int skip_zero_weights) {
    hipLaunchKernelGGL(k_farn_blur_v<float>, grid64x4(W, dst_h, n_frames), dim3(256), 0, s, frames, frame_stride,
                       (long long)pitch0, W, H, pitch0, dst_h, ify, ker_half, half, tmpv, tmpv_frame_stride, skip_zero_weights);
}

void farn_launch_blur_v_u8(hipStream_t s, const unsigned char *frames, long long frame_stride, long long src_pitch,
                           int n_frames, int W, int H, int pitch0, int dst_h, float ify, const float *ker_half, int half,
                           float *tmpv, long long tmpv_frame_stride, int skip_zero_weights) {
    hipLaunchKernelGGL(k_farn_blur_v<unsigned char>, grid64x4(W, dst_h, n_frames), dim3(256), 0, s, frames, frame_stride,
                       src_pitch, W, H, pitch0, dst_h, ify, ker_half, half, tmpv, tmpv_frame_stride, skip_zero_weights);
}

void farn_launch_blur_h_resize(hipStream_t s, const float *tmpv, long long tmpv_frame_stride, int n_frames, int W,
                               int H, int pitch0, int dst_w, int dst_h, int dst_pitch, float ifx, float ify,
                               const float *ker_half, int half, float *pyr, long long pyr_frame_stride,
                               int skip_zero_weights) {
    hipLaunchKernelGGL(k_farn_blur_h_resize, grid64x4(dst_w, dst_h, n_frames), dim3(256), 0, s, tmpv,
                       tmpv_frame_stride, W, H, pitch0, dst_w, dst_h, dst_pitch, ifx, ify, ker_half, half, pyr,
                       pyr_frame_stride, skip_zero_weights);
}

void farn_launch_polyexp(hipStream_t s, const float *pyr, long long pyr_frame_stride, int n_frames,
                         const int *frame_slots, float *frame_R, long long frame_stride, FarnLevelGeom L,
                         FarnPolyConsts pc, int rows) {
    // rows: 16 = k_farn_polyexp_rows<16>, anything else = one image row per workgroup (the first form)
    if (rows == 16) {
        const dim3 grid((L.w + 245) / 246, (L.h + 15) / 16, n_frames);
        hipLaunchKernelGGL(k_farn_polyexp_rows<16>, grid, dim3(256), 0, s, pyr, pyr_frame_stride, frame_slots, frame_R,
                           frame_stride, L, pc);
        return;
    }
    const dim3 grid((L.w + 245) / 246, L.h, n_frames);
    hipLaunchKernelGGL(k_farn_polyexp, grid, dim3(256), 0, s, pyr, pyr_frame_stride, frame_slots, frame_R,
                       frame_stride, L, pc);
}

void farn_launch_init_flow(hipStream_t s, const FarnPairCtx &c, int cur_set, int prev_w, int prev_h, int prev_pitch,
                           float ifx, float ify, float up, int zero) {
    hipLaunchKernelGGL(k_farn_init_flow, grid64x4(c.L.w, c.L.h, c.n_pairs), dim3(256), 0, s, c, cur_set, prev_w,
                       prev_h, prev_pitch, ifx, ify, up, zero);
}

void farn_launch_update_matrices(hipStream_t s, const FarnPairCtx &c, int flow_set, int m_set) {
    hipLaunchKernelGGL(k_farn_update_matrices, grid64x4(c.L.w, c.L.h, c.n_pairs), dim3(256), 0, s, c, flow_set,
                       m_set);
}

void farn_launch_iteration(hipStream_t s, const FarnPairCtx &c, int flow_set, int m_src, int half, float box_inv,
                           int do_matrices, int impl) {
    if (impl == 0 && half == 6) { // winSize 13, the reference's value: the tuned instantiation
        const dim3 grid((c.L.w + 63) / 64, (c.L.h + 31) / 32, c.n_pairs);
        hipLaunchKernelGGL(k_farn_iteration_t<6>, grid, dim3(256), 0, s, c, flow_set, m_src, box_inv, do_matrices);
        return;
    }
    const dim3 grid((c.L.w + 63) / 64, (c.L.h + 15) / 16, c.n_pairs);
    hipLaunchKernelGGL(k_farn_iteration, grid, dim3(256), 0, s, c, flow_set, m_src, half, box_inv, do_matrices);
}

void farn_launch_iter_stream(hipStream_t s, const FarnPairCtx &c, int flow_in, int flow_out, float box_inv, float *merged,
                             long long merged_stride) {
    const int seg_rows = farn_stream_seg_rows(c.L.w, c.L.h, c.n_pairs);
    const dim3 grid((c.L.w + 63) / 64, (c.L.h + seg_rows - 1) / seg_rows, c.n_pairs);
    hipLaunchKernelGGL((k_farn_iter_stream<6, false>), grid, dim3(256), 0, s, c, flow_in, flow_out, box_inv, seg_rows, merged,
                       merged_stride, FarnInit{});
}

void farn_launch_iter_stream_init(hipStream_t s, const FarnPairCtx &c, int prev_set, int flow_out, float box_inv,
                                  float *merged, long long merged_stride, int prev_w, int prev_h, int prev_pitch, float ifx,
                                  float ify, float up, int zero) {
    const int seg_rows = farn_stream_seg_rows(c.L.w, c.L.h, c.n_pairs);
    const dim3 grid((c.L.w + 63) / 64, (c.L.h + seg_rows - 1) / seg_rows, c.n_pairs);
    FarnInit in;
    in.zero = zero, in.prev_w = prev_w, in.prev_h = prev_h, in.prev_pitch = prev_pitch, in.ifx = ifx, in.ify = ify, in.up = up;
    hipLaunchKernelGGL((k_farn_iter_stream<6, true>), grid, dim3(256), 0, s, c, prev_set, flow_out, box_inv, seg_rows, merged,
                       merged_stride, in);
}

void farn_launch_merge(hipStream_t s, const FarnPairCtx &c, int flow_set, float *out, long long out_stride) {
    hipLaunchKernelGGL(k_farn_merge, grid64x4(c.L.w, c.L.h, c.n_pairs), dim3(256), 0, s, c, flow_set, out,
                       out_stride);
}
