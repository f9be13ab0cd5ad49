// brox_kernels.hip — hand-written gfx950 kernels for the -a=brox path.
//
// Replaces cv::cuda::BroxOpticalFlow::calc as the reference calls it (/root/reference/src/denseflow_gpu.cpp:303,
// :332-334): the NCVBroxOpticalFlow kernels (derivative row/column filters, prepare_sor_stage_1_tex,
// prepare_sor_stage_2, sor_pass<red/black>, add, resize, scale) plus the 1/255 convertTo.
// The algorithm is the one DEFINED in oracle/brox_oracle.h (upstream details are only partly known:
// SURVEY.md Appendix C, confidence LOW; reference parity is unpinned).  Every expression below is written in
// the oracle's operation order and the file is compiled with -ffp-contract=off, so results are bit-identical
// to the oracle.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "brox_kernels.h"
#include "tvl1_math_pk.h" // f2, pk_fma, the exact Newton division (shared with the TVL1 kernels)

#define BROX_EPS2 1e-6f
#ifndef BROX_SOR_DEBUG
#define BROX_SOR_DEBUG 0
#endif

static inline dim3 bgrid(int w, int h, int z) { return dim3((w + 63) / 64, (h + 3) / 4, z); }

__device__ __forceinline__ int mirror_idx_any(int i, int n) { // edge-duplicating mirror, any offset
    const int p = 2 * n;
    int m = i % p;
    if (m < 0)
        m += p;
    return m < n ? m : p - 1 - m;
}
// The same for i in [-n, 2n) — one reflection: -1 - i (= ~i) below 0, 2n - 1 - i from n on.  Every index a flow smaller
// than the image can produce is in that range; the general form's 32-bit `%` by a run-time divisor is ~35 VALU
// instructions, and four of them per pixel were a third of stage 1's instruction count (round 5).
__device__ __forceinline__ int mirror_idx_near(int i, int n) {
    const int r = i ^ (i >> 31);
    return r < n ? r : 2 * n - 1 - r;
}
__device__ __forceinline__ int mirror_idx(int i, int n) {
    return __builtin_expect(i < -n || i >= 2 * n, 0) ? mirror_idx_any(i, n) : mirror_idx_near(i, n);
}
// 1.0f / sqrtf(s), both IEEE, for s in [2^-85, 2^127) (the callers add BROX_EPS2 = 1e-6 to a sum of squares): the square
// root is the core sequence of tvl1_sqrt_scaled (correctly rounded for every float of that range: checked exhaustively on
// the device, tests/test_device_math_gpu.py) without its scaling, the reciprocal the exact Newton division of tvl1_math.h
// (checked against IEEE division over the Brox range of denominators) — 16 instructions instead of the compiler's ~27 for
// the two library expansions with their range scaling and fix-ups.
__device__ __forceinline__ float inv_sqrtf_ieee(float s) {
    const float y = __builtin_amdgcn_rsqf(s);
    float g = s * y;
    float h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, r, h);
    g = __builtin_fmaf(g, r, g);
    const float d = __builtin_fmaf(-g, g, s);
    g = __builtin_fmaf(d, h, g);
    return tvl1_div(1.0f, g);
}
__device__ __forceinline__ float brox_bicubic_w(float x_) {
    const float x = fabsf(x_);
    const float near = x * x * (1.5f * x - 2.5f) + 1.0f;
    const float far = x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
    return x <= 1.0f ? near : (x < 2.0f ? far : 0.0f);
}
__device__ __forceinline__ float *bplane(const BroxLevelCtx &c, int pair, int plane) {
    return c.planes + (long long)pair * c.slot_stride + (long long)plane * c.plane_stride;
}
__device__ __forceinline__ int du_plane(int d_set) { return d_set ? BROX_PL_DU1 : BROX_PL_DU; }
__device__ __forceinline__ int dv_plane(int d_set) { return d_set ? BROX_PL_DV1 : BROX_PL_DV; }
__device__ __forceinline__ const float *fplane(const BroxLevelCtx &c, int slot, int plane) {
    return c.frames + (long long)slot * c.frame_stride + (long long)plane * c.pyr_elems + c.lvl_off;
}

// ------------------------------------------------------------------------------------------------ frames

__global__ __launch_bounds__(256) void k_brox_u8_to_f32(const unsigned char *src, long long src_frame_stride,
                                                        long long src_pitch, const int *frame_slots, float *frames,
                                                        long long frame_stride, int w, int h, int pitch, float scale) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    const int z = blockIdx.z;
    frames[(long long)frame_slots[z] * frame_stride + (long long)y * pitch + x] =
        (float)src[(long long)z * src_frame_stride + (long long)y * src_pitch + x] * scale;
}

// area-averaging ("supersample") down-sampling of plane I from level l-1 to level l
__global__ __launch_bounds__(256) void k_brox_downsample(float *frames, long long frame_stride, const int *frame_slots,
                                                         long long src_off, int sw, int sh, int spitch,
                                                         long long dst_off, int dw, int dh, int dpitch, float factor) {
    const DfxBlockXY blk = dfx_block_xy(); // XCD-aware (dfx_device.h): neighbouring rows share one L2
    const int ix = blk.x * 64 + (threadIdx.x & 63);
    const int iy = blk.y * 4 + (threadIdx.x >> 6);
    if (ix >= dw || iy >= dh)
        return;
    float *base = frames + (long long)frame_slots[blockIdx.z] * frame_stride;
    const float *src = base + src_off;
    const float s = 1.0f / factor;
    const float y = s * (float)iy, x = s * (float)ix;
    const int yb = (int)floorf(y), ye = (int)ceilf(y + s);
    const int xb = (int)floorf(x), xe = (int)ceilf(x + s);
    float sum = 0.f, wsum = 0.f;
    for (int cy = yb; cy < ye; ++cy) {
        const float wy = fminf((float)cy + 1.0f, y + s) - fmaxf((float)cy, y);
        const float *row = src + (long long)min(cy, sh - 1) * spitch;
        for (int cx = xb; cx < xe; ++cx) {
            const float wx = fminf((float)cx + 1.0f, x + s) - fmaxf((float)cx, x);
            const float w = wx * wy;
            sum = sum + w * row[min(cx, sw - 1)];
            wsum = wsum + w;
        }
    }
    base[dst_off + (long long)iy * dpitch + ix] = sum / wsum;
}

__global__ __launch_bounds__(256) void k_brox_deriv(float *frames, long long frame_stride, const int *frame_slots,
                                                    long long src_off, long long dst_off, int w, int h, int pitch,
                                                    int axis) {
    const DfxBlockXY blk = dfx_block_xy(); // XCD-aware (dfx_device.h): neighbouring rows share one L2
    const int x = blk.x * 64 + (threadIdx.x & 63);
    const int y = blk.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    float *base = frames + (long long)frame_slots[blockIdx.z] * frame_stride;
    const float *src = base + src_off;
    const float kD[5] = {1.0f, -8.0f, 0.0f, 8.0f, -1.0f};
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        // taps at distance <= 2 of a pixel: inside [-n, 2n) whenever n >= 2 (a wave-uniform test; Brox levels are >= 16 px)
        const int tx = x + k - 2, ty = y + k - 2;
        const float v = axis == 0 ? src[(long long)y * pitch + (w >= 2 ? mirror_idx_near(tx, w) : mirror_idx_any(tx, w))]
                                  : src[(long long)(h >= 2 ? mirror_idx_near(ty, h) : mirror_idx_any(ty, h)) * pitch + x];
        s = s + v * kD[k];
    }
    base[dst_off + (long long)y * pitch + x] = s * (1.0f / 12.0f);
}

// ------------------------------------------------------------------------------------------------ per pair

__global__ __launch_bounds__(256) void k_brox_level_init(BroxLevelCtx c, int uv_set, int zero_uv) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= c.w || y >= c.h)
        return;
    const int b = blockIdx.z;
    const long long o = (long long)y * c.pitch + x;
    bplane(c, b, BROX_PL_DU)[o] = 0.0f;
    bplane(c, b, BROX_PL_DV)[o] = 0.0f;
    if (zero_uv) {
        bplane(c, b, BROX_PL_U0 + 2 * uv_set)[o] = 0.0f;
        bplane(c, b, BROX_PL_V0 + 2 * uv_set)[o] = 0.0f;
    }
}

// Loads and stores at a 32-bit BYTE offset from a wave-uniform plane pointer: the instruction then takes the pointer from
// scalar registers and the offset from one VGPR (global_load_dword v, v_off, s[base:base+1]).  With an ELEMENT offset the
// byte offset is a 34-bit quantity as far as the compiler can tell, and every access pays a 64-bit shift and a 64-bit add on
// the vector ALU (42 of stage 1's 437 vector instructions, round 6; 12 remain: where several planes are read at ONE offset the
// optimiser forms one vector pointer and adds the planes' strides to it).  A level's plane is < 2^23 elements (3840 x 2160).
typedef __attribute__((address_space(1))) char brox_glb_char;
__device__ __forceinline__ brox_glb_char *opaque_base(const void *p) { // a wave-uniform pointer the optimiser cannot relate to another
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (brox_glb_char *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float ld_at(const float *p, unsigned byte_off) {
    return *(const __attribute__((address_space(1))) float *)(opaque_base(p) + byte_off);
}
__device__ __forceinline__ void st_at(float *p, unsigned byte_off, float v) {
    *(__attribute__((address_space(1))) float *)(opaque_base(p) + byte_off) = v;
}
struct BlTap {                   // byte offsets inside a level's plane (see ld_at)
    unsigned i00, i01, i10, i11;
    float ax, ay;
};
__device__ __forceinline__ BlTap bl_setup(float fx, float fy, int w, int h, int pitch) {
    BlTap t;
    fx = fminf(fmaxf(fx, -1.0e6f), 1.0e6f);
    fy = fminf(fmaxf(fy, -1.0e6f), 1.0e6f);
    const float x0 = floorf(fx), y0 = floorf(fy);
    t.ax = fx - x0;
    t.ay = fy - y0;
    const int ix = (int)x0, iy = (int)y0;
    int xa, xb, ya, yb;
    if (__builtin_expect(ix < -w || ix + 1 >= 2 * w || iy < -h || iy + 1 >= 2 * h, 0)) { // a flow larger than the image
        xa = mirror_idx_any(ix, w), xb = mirror_idx_any(ix + 1, w);
        ya = mirror_idx_any(iy, h), yb = mirror_idx_any(iy + 1, h);
    } else { // one branch for the four indices
        xa = mirror_idx_near(ix, w), xb = mirror_idx_near(ix + 1, w);
        ya = mirror_idx_near(iy, h), yb = mirror_idx_near(iy + 1, h);
    }
    t.i00 = 4u * (unsigned)(ya * pitch + xa);
    t.i01 = 4u * (unsigned)(ya * pitch + xb);
    t.i10 = 4u * (unsigned)(yb * pitch + xa);
    t.i11 = 4u * (unsigned)(yb * pitch + xb);
    return t;
}
__device__ __forceinline__ float bl_sample(const float *p, const BlTap &t) {
    const float a = (1.0f - t.ax) * ld_at(p, t.i00) + t.ax * ld_at(p, t.i01);
    const float b = (1.0f - t.ax) * ld_at(p, t.i10) + t.ax * ld_at(p, t.i11);
    return (1.0f - t.ay) * a + t.ay * b;
}
// two planes at once: the same three rounded operations per half (v_pk_mul / v_pk_add), same bits as bl_sample
__device__ __forceinline__ f2 bl_sample2(const float *p, const float *q, const BlTap &t) {
    const f2 wx0 = (f2)(1.0f - t.ax), wx1 = (f2)(t.ax), wy0 = (f2)(1.0f - t.ay), wy1 = (f2)(t.ay);
    const f2 a = wx0 * pk_set(ld_at(p, t.i00), ld_at(q, t.i00)) + wx1 * pk_set(ld_at(p, t.i01), ld_at(q, t.i01));
    const f2 b = wx0 * pk_set(ld_at(p, t.i10), ld_at(q, t.i10)) + wx1 * pk_set(ld_at(p, t.i11), ld_at(q, t.i11));
    return wy0 * a + wy1 * b;
}

// stage 1: data-term coefficients and staggered diffusivities (brox_oracle.h, step 3b).
// The diffusivities read w = u + du and v + dv at 14 neighbour positions per pixel; the workgroup builds both
// sums once for its 64 x 4 pixels plus a 1-pixel ring (coordinates clamped to the image, which is exactly the
// max(x-1, 0) / min(x+1, w-1) addressing of the definition) in LDS instead of 56 global loads per pixel.
__global__ __launch_bounds__(256) void k_brox_stage1(BroxLevelCtx c, int uv_set, int d_set) {
    __shared__ float WUs[6][68], WVs[6][68];
    const DfxBlockXY blk = dfx_block_xy(); // 1-pixel ring + warped samples shared with the rows above / below: one L2
    const int x0 = blk.x * 64, y0 = blk.y * 4;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    const int b = blockIdx.z, w = c.w, h = c.h, pitch = c.pitch;
    const float *u = bplane(c, b, BROX_PL_U0 + 2 * uv_set), *v = bplane(c, b, BROX_PL_V0 + 2 * uv_set);
    const float *DU = bplane(c, b, du_plane(d_set)), *DV = bplane(c, b, dv_plane(d_set));
    // the tile: every thread its own pixel (coordinates clamped to the image: what a neighbour inside the image reads there),
    // threads 0 .. 139 one entry of the 1-pixel ring each (rows 0 and 5: 2 x 66, columns 0 and 65 of rows 1 .. 4: 8)
    const unsigned oc = 4u * (unsigned)(min(y, h - 1) * pitch + min(x, w - 1));
    const float u_c = ld_at(u, oc), v_c = ld_at(v, oc), du_c = ld_at(DU, oc), dv_c = ld_at(DV, oc);
    WUs[ly + 1][lx + 1] = u_c + du_c;
    WVs[ly + 1][lx + 1] = v_c + dv_c;
    if (threadIdx.x < 140) {
        const int e = (int)threadIdx.x, k = e - 132;
        const int ty = e < 66 ? 0 : (e < 132 ? 5 : 1 + (k & 3));
        const int tx = e < 66 ? e : (e < 132 ? e - 66 : (k < 4 ? 0 : 65));
        const unsigned so = 4u * (unsigned)(min(max(y0 - 1 + ty, 0), h - 1) * pitch + min(max(x0 - 1 + tx, 0), w - 1));
        WUs[ty][tx] = ld_at(u, so) + ld_at(DU, so);
        WVs[ty][tx] = ld_at(v, so) + ld_at(DV, so);
    }
    __syncthreads();
    if (x >= w || y >= h)
        return;
    const PairDesc pd = c.pairs[b];
    const unsigned o = oc; // byte offset inside the level's plane (ld_at / st_at): inside the image oc is the pixel's own
    // tile coordinates of (x, y) are (lx + 1, ly + 1); m / p = the clamped neighbours
#define WU(dx, dy) WUs[ly + 1 + (dy)][lx + 1 + (dx)]
#define WV(dx, dy) WVs[ly + 1 + (dy)][lx + 1 + (dx)]
    const BlTap t = bl_setup((float)x + u_c, (float)y + v_c, w, h, pitch);
    const f2 s01 = bl_sample2(fplane(c, pd.frame_b, BROX_FP_I), fplane(c, pd.frame_b, BROX_FP_DX), t);
    const f2 s23 = bl_sample2(fplane(c, pd.frame_b, BROX_FP_DY), fplane(c, pd.frame_b, BROX_FP_DXX), t);
    const f2 s45 = bl_sample2(fplane(c, pd.frame_b, BROX_FP_DXY), fplane(c, pd.frame_b, BROX_FP_DYY), t);
    const float I1w = s01.x, Ixw = s01.y, Iyw = s23.x, Ixxw = s23.y, Ixyw = s45.x, Iyyw = s45.y;
    const float Iz = I1w - ld_at(fplane(c, pd.frame_a, BROX_FP_I), o);
    const float Ixz = Ixw - ld_at(fplane(c, pd.frame_a, BROX_FP_DX), o);
    const float Iyz = Iyw - ld_at(fplane(c, pd.frame_a, BROX_FP_DY), o);
    const float du = du_c, dv = dv_c;
    const float gamma = c.gamma;
    const float q0 = Iz + (Ixw * du + Iyw * dv);
    const float q1 = Ixz + (Ixxw * du + Ixyw * dv);
    const float q2 = Iyz + (Ixyw * du + Iyyw * dv);
    const float psi = (0.5f * inv_sqrtf_ieee((q0 * q0 + gamma * (q1 * q1 + q2 * q2)) + BROX_EPS2)) / c.alpha;
    st_at(bplane(c, b, BROX_PL_NDUDV), o, psi * (Ixw * Iyw + gamma * (Ixxw * Ixyw + Ixyw * Iyyw)));
    st_at(bplane(c, b, BROX_PL_IDU), o, psi * (Ixw * Ixw + gamma * (Ixyw * Ixyw + Ixxw * Ixxw)));
    st_at(bplane(c, b, BROX_PL_IDV), o, psi * (Iyw * Iyw + gamma * (Ixyw * Ixyw + Iyyw * Iyyw)));
    st_at(bplane(c, b, BROX_PL_NU), o, psi * (Ixw * Iz + gamma * (Ixxw * Ixz + Ixyw * Iyz)));
    st_at(bplane(c, b, BROX_PL_NV), o, psi * (Iyw * Iz + gamma * (Iyyw * Iyz + Ixyw * Ixz)));
    float gx = 0.0f, gy = 0.0f;
    if (x > 0) {
        const float ux = WU(0, 0) - WU(-1, 0), vx = WV(0, 0) - WV(-1, 0);
        const float uy = 0.25f * (((WU(0, 1) + WU(-1, 1)) - WU(0, -1)) - WU(-1, -1));
        const float vy = 0.25f * (((WV(0, 1) + WV(-1, 1)) - WV(0, -1)) - WV(-1, -1));
        gx = 0.5f * inv_sqrtf_ieee((((ux * ux + uy * uy) + vx * vx) + vy * vy) + BROX_EPS2);
    }
    if (y > 0) {
        const float uy = WU(0, 0) - WU(0, -1), vy = WV(0, 0) - WV(0, -1);
        const float ux = 0.25f * (((WU(1, 0) + WU(1, -1)) - WU(-1, 0)) - WU(-1, -1));
        const float vx = 0.25f * (((WV(1, 0) + WV(1, -1)) - WV(-1, 0)) - WV(-1, -1));
        gy = 0.5f * inv_sqrtf_ieee((((ux * ux + uy * uy) + vx * vx) + vy * vy) + BROX_EPS2);
    }
    st_at(bplane(c, b, BROX_PL_GX), o, gx);
    st_at(bplane(c, b, BROX_PL_GY), o, gy);
#undef WU
#undef WV
}

__global__ __launch_bounds__(256) void k_brox_stage2(BroxLevelCtx c) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= c.w || y >= c.h)
        return;
    const int b = blockIdx.z;
    const long long o = (long long)y * c.pitch + x;
    const float *GX = bplane(c, b, BROX_PL_GX), *GY = bplane(c, b, BROX_PL_GY);
    const float gl = GX[o], gr = (x + 1 < c.w) ? GX[o + 1] : 0.0f;
    const float gd = GY[o], gu = (y + 1 < c.h) ? GY[o + c.pitch] : 0.0f;
    const float gs = ((gl + gr) + gd) + gu;
    float *IDU = bplane(c, b, BROX_PL_IDU), *IDV = bplane(c, b, BROX_PL_IDV);
    IDU[o] = 1.0f / (IDU[o] + gs);
    IDV[o] = 1.0f / (IDV[o] + gs);
}

// one red/black SOR half-sweep: only pixels with (x + y) % 2 == color are updated (in place; their four
// neighbours have the other colour and are not touched by this launch)
__global__ __launch_bounds__(256) void k_brox_sor(BroxLevelCtx c, int uv_set, int d_set, int color) {
    const int xh = blockIdx.x * 64 + (threadIdx.x & 63); // half-resolution column index
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= c.h)
        return;
    const int x = 2 * xh + ((y + color) & 1);
    if (x >= c.w)
        return;
    const int b = blockIdx.z, w = c.w, h = c.h, pitch = c.pitch;
    const float *u = bplane(c, b, BROX_PL_U0 + 2 * uv_set), *v = bplane(c, b, BROX_PL_V0 + 2 * uv_set);
    float *DU = bplane(c, b, du_plane(d_set)), *DV = bplane(c, b, dv_plane(d_set));
    const float *GX = bplane(c, b, BROX_PL_GX), *GY = bplane(c, b, BROX_PL_GY);
    const long long o = (long long)y * pitch + x;
    const long long ol = (long long)y * pitch + max(x - 1, 0), orr = (long long)y * pitch + min(x + 1, w - 1);
    const long long od = (long long)max(y - 1, 0) * pitch + x, ou = (long long)min(y + 1, h - 1) * pitch + x;
    const float gl = GX[o], gr = (x + 1 < w) ? GX[o + 1] : 0.0f;
    const float gd = GY[o], gu = (y + 1 < h) ? GY[o + pitch] : 0.0f;
    const float gs = ((gl + gr) + gd) + gu;
    const float su =
        (((gl * (u[ol] + DU[ol]) + gr * (u[orr] + DU[orr])) + gd * (u[od] + DU[od])) + gu * (u[ou] + DU[ou])) - gs * u[o];
    const float sv =
        (((gl * (v[ol] + DV[ol]) + gr * (v[orr] + DV[orr])) + gd * (v[od] + DV[od])) + gu * (v[ou] + DV[ou])) - gs * v[o];
    const float du = DU[o], dv = DV[o];
    const float ndudv = bplane(c, b, BROX_PL_NDUDV)[o];
    const float omega = c.omega;
    const float du_n =
        (1.0f - omega) * du + omega * (bplane(c, b, BROX_PL_IDU)[o] * ((su - bplane(c, b, BROX_PL_NU)[o]) - ndudv * dv));
    const float dv_n =
        (1.0f - omega) * dv + omega * (bplane(c, b, BROX_PL_IDV)[o] * ((sv - bplane(c, b, BROX_PL_NV)[o]) - ndudv * du_n));
    DU[o] = du_n;
    DV[o] = dv_n;
}

// ------------------------------------------------------------------------------------------------
// The fused SOR (the tuned default): S full red+black sweeps per launch on a 64 x 64 LDS tile, one thread per 2-column x
// 2-row patch, the 9 per-pixel coefficients and u, v in registers, only w = u + du (what neighbours read) in LDS.  Tile
// origins are even, so a pixel's colour is a compile-time function of its position in the patch.  Every half sweep
// shrinks the valid region by one pixel: a halo of 2*S pixels is recomputed redundantly and only the inner region is
// written back (to the OTHER du / dv set: with an in-place update neighbouring workgroups of one launch race on each
// other's halo).  Same per-pixel expression, same order: bit-identical to k_brox_sor, the one-launch-per-half-sweep form.
// One barrier per half sweep: a half sweep of colour c reads w only at pixels of colour 1-c (the four neighbours; at a
// clamped tile edge the pixel's own entry) and writes w only at pixels of colour c, each by the thread that owns it.
// Round 3 rebuilt the round-2 kernel (VALU busy 33 %, 63 % of wave cycles waiting: profiles/round2/brox/) around:
//   * LOAD PHASE.  The round-2 kernel fetched each of its 15 per-pixel inputs with a dword load whose lanes are 8 bytes
//     apart (x = 2 * lane + k): 60 load instructions per thread, each using half of the bytes the texture-address unit
//     walks.  Here a patch row is one 8-byte load per plane (x0, lx0 are even and the pitch is a multiple of 64, so the
//     pair is 8-byte aligned): 29 load instructions, every byte used.  gr = GX(x+1) and gu = GY(y+1) of the patch's
//     inner edges are the neighbour's gl / gd, already in registers.
//   * SWEEPS.  The two pixels a thread updates in a half sweep (one colour = one diagonal of the patch) are independent:
//     the 14 per-pixel values are held as float2 {diagonal element 0, element 1} per colour and the 34 float operations
//     of an update run once as v_pk_* for both pixels (IEEE per half, nothing contracted: same bits).
//   * the two reciprocals of stage 2 use the exact Newton sequence of tvl1_math.h (checked element-wise on the GPU
//     against IEEE division over the Brox range of denominators, tests/test_device_math_gpu.py) on both pixels at once.
typedef float f2_a8 __attribute__((ext_vector_type(2), aligned(8)));

// SYNC (round 6): how a half sweep waits for the one before it.
//   0  __syncthreads() (the default): every half sweep of the tile is one phase of all sixteen waves — they read LDS
//      together, run their dependent update chains together and meet at the barrier together: ~1400 cycles per half sweep
//      for ~620 VALU and ~640 LDS cycles that never overlap (profiles/round3/brox/);
//   1  band by band (DFX_VAR_BROX_SOR_PROGRESS): a wave owns 4 tile rows, and half sweep t of a band depends only on half
//      sweep t - 1 of the bands above and below it (colour c reads colour 1 - c at the four neighbours and writes colour c:
//      a read-after-write and a write-after-read on the neighbour bands' edge rows, both settled once THOSE bands have
//      finished t - 1).  Each wave publishes its progress in LDS (release) and waits for its two neighbours' (acquire): no
//      workgroup barrier inside the sweeps.  Same updates, same order per pixel: bit-identical — and 7 % SLOWER (232 vs
//      249 pairs/s at 1080p, 59.2 vs 63.5 at 4K -s=2; profiles/round6/brox/): because every band needs BOTH neighbours'
//      previous half sweep, no band can run ahead of the one next to it at all — the dependency graph has the barrier's
//      shape, only its implementation got more expensive (two polled LDS words and a release fence per wave and half
//      sweep).  What would decouple the phases is a second, independent tile on the same CU, and the register file has no
//      room for one (a 64 x 64 tile's 14 values per pixel are 224 KB of its 512 KB): DESIGN.md section 4.
template <int S, int SYNC>
__global__ __launch_bounds__(1024) void k_brox_sor_pk(BroxLevelCtx c, int uv_set, int d_src, int n_sweeps, int tiles_x) {
    constexpr int TW = 64, TH = 64, HALO = 2 * S;
    // w = u + du and v + dv of the tile, split by column parity: W*[x & 1][y][x >> 1].  A lane owns columns 2 * pcol and
    // 2 * pcol + 1, so in a row-major [y][x] tile every access of a wave has a lane stride of TWO floats — a two-way
    // bank conflict on all 16 reads and 4 writes of a half sweep (ds_read_b32 banks are (a / 4) mod 32,
    // MI355X_MICROARCH.md), and those LDS cycles, not the arithmetic, were the sweeps' time.  Split by parity every
    // access is lane-consecutive: pixel (y, 2p + k) is W*[k][y][p], its horizontal neighbours W*[1 - k][y][p] and
    // W*[1 - k][y][p -/+ 1].
    __shared__ float WU[2][TH][TW / 2];
    __shared__ float WV[2][TH][TW / 2];
    __shared__ int progress[TH / 4]; // SYNC = 1: half sweeps completed per band
#define WU_AT(ly, lx) WU[(lx)&1][ly][(lx) >> 1]
#define WV_AT(ly, lx) WV[(lx)&1][ly][(lx) >> 1]
    const int b = blockIdx.z, w = c.w, h = c.h, pitch = c.pitch;
    const int tile = dfx_block_linear(); // neighbouring tiles re-read each other's 2*S-pixel halo: keep them in one L2
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * (TW - 2 * HALO) - HALO; // even
    const int y0 = ty * (TH - 2 * HALO) - HALO; // even
    const int pcol = threadIdx.x & 31, prow = threadIdx.x >> 5;
    const int lx0 = 2 * pcol, ly0 = 2 * prow;
    const int x = x0 + lx0, y = y0 + ly0; // pixel (row 0, column 0) of the patch

    const float *u = bplane(c, b, BROX_PL_U0 + 2 * uv_set), *v = bplane(c, b, BROX_PL_V0 + 2 * uv_set);
    const float *DU = bplane(c, b, du_plane(d_src)), *DV = bplane(c, b, dv_plane(d_src));
    float *DUo = bplane(c, b, du_plane(d_src ^ 1)), *DVo = bplane(c, b, dv_plane(d_src ^ 1));
    const float *GX = bplane(c, b, BROX_PL_GX), *GY = bplane(c, b, BROX_PL_GY);
    const float *IDU = bplane(c, b, BROX_PL_IDU), *IDV = bplane(c, b, BROX_PL_IDV);
    const float *NDUDV = bplane(c, b, BROX_PL_NDUDV), *NU = bplane(c, b, BROX_PL_NU), *NV = bplane(c, b, BROX_PL_NV);

    // ---- load phase: rows i = 0, 1 of the patch as float2 (k = 0, 1), masked to 0 outside the image
    const bool inx0 = x >= 0 && x < w, inx1 = x + 1 >= 0 && x + 1 < w; // x is even: x < 0 puts both columns outside
    const bool iny[3] = {y >= 0 && y < h, y + 1 >= 0 && y + 1 < h, y + 2 >= 0 && y + 2 < h};
    long long o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#if BROX_SOR_DEBUG == 2 // measurement build only: the sweeps without the HBM traffic of the load phase (WRONG flows)
        o[i] = 2 * (threadIdx.x & 31);
#else
        o[i] = (inx0 && iny[i]) ? ((long long)(y + i) * pitch + x) : 0; // masked rows read element 0
#endif
    auto ld2 = [](const float *P, long long off) -> f2 { return *reinterpret_cast<const f2_a8 *>(P + off); };
    auto mask = [&](f2 r, int i) -> f2 { return pk_set(inx0 && iny[i] ? r.x : 0.0f, inx1 && iny[i] ? r.y : 0.0f); };
    f2 r_gl[2], r_gd[2], r_idu[2], r_idv[2], r_nd[2], r_nu[2], r_nv[2], r_u[2], r_v[2], r_du[2], r_dv[2];
    float gxr[2]; // GX at x + 2: the right neighbour's diffusivity of column 1
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        r_gl[i] = ld2(GX, o[i]);
        r_gd[i] = ld2(GY, o[i]);
        r_idu[i] = ld2(IDU, o[i]);
        r_idv[i] = ld2(IDV, o[i]);
        r_nd[i] = ld2(NDUDV, o[i]);
        r_nu[i] = ld2(NU, o[i]);
        r_nv[i] = ld2(NV, o[i]);
        r_u[i] = ld2(u, o[i]);
        r_v[i] = ld2(v, o[i]);
        r_du[i] = ld2(DU, o[i]);
        r_dv[i] = ld2(DV, o[i]);
        gxr[i] = GX[(x + 2 < w) ? o[i] + 2 : o[i]];
    }
    const f2 r_gy2 = ld2(GY, o[2]);
    f2 r_gr[2], r_gu[2], r_gs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        r_gl[i] = mask(r_gl[i], i);
        r_gd[i] = mask(r_gd[i], i);
        r_nd[i] = mask(r_nd[i], i);
        r_nu[i] = mask(r_nu[i], i);
        r_nv[i] = mask(r_nv[i], i);
        r_u[i] = mask(r_u[i], i);
        r_v[i] = mask(r_v[i], i);
        r_du[i] = mask(r_du[i], i);
        r_dv[i] = mask(r_dv[i], i);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // gr = (in && x + 1 < w) ? GX[o + 1] : 0.  Column 0: GX[o + 1] is column 1's own (masked) gl.
        r_gr[i] = pk_set(inx0 && iny[i] ? r_gl[i].y : 0.0f, (inx1 && iny[i] && x + 2 < w) ? gxr[i] : 0.0f);
        // gu = (in && y + 1 < h) ? GY[o + pitch] : 0.  Row 0: the (masked) gd of row 1; row 1: row 2 of GY.
        if (i == 0)
            r_gu[i] = pk_set(inx0 && iny[0] ? r_gd[1].x : 0.0f, inx1 && iny[0] ? r_gd[1].y : 0.0f);
        else
            r_gu[i] = pk_set(inx0 && iny[1] && iny[2] ? r_gy2.x : 0.0f, inx1 && iny[1] && iny[2] ? r_gy2.y : 0.0f);
        r_gs[i] = ((r_gl[i] + r_gr[i]) + r_gd[i]) + r_gu[i];
        // stage 2 of the definition, 1 / (data term + sum of diffusivities): the exact division, two pixels at once
        const f2 den_u = r_idu[i] + r_gs[i], den_v = r_idv[i] + r_gs[i];
        r_idu[i] = mask(pk_div_with_rcp((f2)(1.0f), den_u, pk_refined_rcp(den_u)), i);
        r_idv[i] = mask(pk_div_with_rcp((f2)(1.0f), den_v, pk_refined_rcp(den_v)), i);
        const f2 wu = r_u[i] + r_du[i], wv = r_v[i] + r_dv[i];
        WU[0][ly0 + i][pcol] = wu.x, WU[1][ly0 + i][pcol] = wu.y;
        WV[0][ly0 + i][pcol] = wv.x, WV[1][ly0 + i][pcol] = wv.y;
    }
    // colour pairs: pair q = {element 0: (row 0, column q), element 1: (row 1, column 1 - q)}; (x + y) parity of a patch
    // element (i, k) is (i + k) & 1 because x0, y0, lx0, ly0 are even, so pair q is what half sweep `q` updates
#define BROX_PAIR(r) {pk_set(r[0].x, r[1].y), pk_set(r[0].y, r[1].x)}
    const f2 gl[2] = BROX_PAIR(r_gl), gr[2] = BROX_PAIR(r_gr), gd[2] = BROX_PAIR(r_gd), gu[2] = BROX_PAIR(r_gu);
    const f2 gs[2] = BROX_PAIR(r_gs), idu[2] = BROX_PAIR(r_idu), idv[2] = BROX_PAIR(r_idv), nd[2] = BROX_PAIR(r_nd);
    const f2 nu[2] = BROX_PAIR(r_nu), nv[2] = BROX_PAIR(r_nv), uu[2] = BROX_PAIR(r_u), vv[2] = BROX_PAIR(r_v);
    f2 du[2] = BROX_PAIR(r_du), dv[2] = BROX_PAIR(r_dv);
#undef BROX_PAIR
    if (SYNC == 1 && threadIdx.x < TH / 4)
        progress[threadIdx.x] = 0;
    __syncthreads();

    // neighbour coordinates, clamped at the tile edge exactly like max(lx - 1, 0) / min(lx + 1, TW - 1) of the definition's
    // tile form (a clamped read returns the pixel's own entry; such pixels are halo and their values are discarded)
    const int xm = max(lx0 - 1, 0), xp = min(lx0 + 2, TW - 1), ym = max(ly0 - 1, 0), yp = min(ly0 + 2, TH - 1);
    const float omega = c.omega, om1 = 1.0f - omega;
    constexpr int ROWS_PER_WAVE = 4;                         // 32 patch columns x 2 patch rows per wavefront
    const int band0 = (ly0 / ROWS_PER_WAVE) * ROWS_PER_WAVE; // first tile row of this wavefront
#if BROX_SOR_DEBUG == 1 // measurement build only (scripts/build_variant.sh): load and store phases without the sweeps (WRONG flows)
    n_sweeps = 0;
#endif
    const int band = ly0 / ROWS_PER_WAVE, lane = threadIdx.x & 63;
    for (int sw = 0; sw < n_sweeps; ++sw) {
        // a wavefront whose rows can no longer influence the owned region skips its updates: after sweep s of n the
        // result is needed on the owned rows +- (2 (n - 1 - s) + 1)
        const int m = 2 * (n_sweeps - 1 - sw) + 1;
        const bool live = band0 + ROWS_PER_WAVE - 1 >= HALO - m && band0 < TH - HALO + m;
        if (SYNC == 1 && !live) { // dead from here on (m only shrinks): nobody must wait for this band again
            if (lane == 0)
                __hip_atomic_store(&progress[band], 1 << 20, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (SYNC == 1) { // the bands above and below have finished half sweep t - 1
                const int t = 2 * sw + q;
                if (band > 0)
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&progress[band - 1], __ATOMIC_ACQUIRE,
                                                                            __HIP_MEMORY_SCOPE_WORKGROUP)) < t)
                        __builtin_amdgcn_s_sleep(1);
                if (band < TH / ROWS_PER_WAVE - 1)
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&progress[band + 1], __ATOMIC_ACQUIRE,
                                                                            __HIP_MEMORY_SCOPE_WORKGROUP)) < t)
                        __builtin_amdgcn_s_sleep(1);
            }
            if (live) {
                // element 0 = (ly0, lx0 + q), element 1 = (ly0 + 1, lx0 + 1 - q)
                const int ax = lx0 + q, bx = lx0 + 1 - q;
                const int axl = q ? lx0 : xm, axr = q ? xp : lx0 + 1; // left / right neighbour columns of element 0
                const int bxl = q ? xm : lx0, bxr = q ? lx0 + 1 : xp; // ... of element 1
                const f2 Lu = pk_set(WU_AT(ly0, axl), WU_AT(ly0 + 1, bxl)), Ru = pk_set(WU_AT(ly0, axr), WU_AT(ly0 + 1, bxr));
                const f2 Du = pk_set(WU_AT(ym, ax), WU_AT(ly0, bx)), Uu = pk_set(WU_AT(ly0 + 1, ax), WU_AT(yp, bx));
                const f2 Lv = pk_set(WV_AT(ly0, axl), WV_AT(ly0 + 1, bxl)), Rv = pk_set(WV_AT(ly0, axr), WV_AT(ly0 + 1, bxr));
                const f2 Dv = pk_set(WV_AT(ym, ax), WV_AT(ly0, bx)), Uv = pk_set(WV_AT(ly0 + 1, ax), WV_AT(yp, bx));
                const f2 su = (((gl[q] * Lu + gr[q] * Ru) + gd[q] * Du) + gu[q] * Uu) - gs[q] * uu[q];
                const f2 sv = (((gl[q] * Lv + gr[q] * Rv) + gd[q] * Dv) + gu[q] * Uv) - gs[q] * vv[q];
                const f2 du_n = om1 * du[q] + omega * (idu[q] * ((su - nu[q]) - nd[q] * dv[q]));
                const f2 dv_n = om1 * dv[q] + omega * (idv[q] * ((sv - nv[q]) - nd[q] * du_n));
                du[q] = du_n;
                dv[q] = dv_n;
                // publish right away: this half sweep reads w only at pixels of the other colour (or, clamped, at the
                // pixel's own entry) and each entry of this colour is written by the one thread that owns it
                const f2 wu = uu[q] + du_n, wv = vv[q] + dv_n;
                WU_AT(ly0, ax) = wu.x;
                WU_AT(ly0 + 1, bx) = wu.y;
                WV_AT(ly0, ax) = wv.x;
                WV_AT(ly0 + 1, bx) = wv.y;
            }
            if (SYNC == 1) { // publish: the release orders the wave's LDS writes (one wait counter per wave) in front of it
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0)
                    __hip_atomic_store(&progress[band], 2 * sw + q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __syncthreads();
            }
        }
    }
    // ---- store the owned region into the other du / dv set
    if (lx0 >= HALO && lx0 < TW - HALO && x < w) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ly = ly0 + i, yy = y + i;
            if (ly >= HALO && ly < TH - HALO && yy < h) {
                const long long oo = (long long)yy * pitch + x;
                const f2 a = i == 0 ? pk_set(du[0].x, du[1].x) : pk_set(du[1].y, du[0].y);
                const f2 bq = i == 0 ? pk_set(dv[0].x, dv[1].x) : pk_set(dv[1].y, dv[0].y);
                if (x + 1 < w) {
                    *reinterpret_cast<f2_a8 *>(DUo + oo) = a;
                    *reinterpret_cast<f2_a8 *>(DVo + oo) = bq;
                } else {
                    DUo[oo] = a.x;
                    DVo[oo] = bq.x;
                }
            }
        }
    }
}
#undef WU_AT
#undef WV_AT

// ------------------------------------------------------------------------------------------------
// The streaming form of the fused SOR (round 6): ONE persistent 1024-thread workgroup per CU walks over tiles, and while it
// sweeps tile i the seven coefficient planes of tile i + 1 (GX, GY, IDU, IDV, NDUDV, NU, NV: 64 x 64 floats each, 112 KB of
// the CU's 160 KB of LDS) arrive by LDS-DMA (global_load_lds: no registers, no VALU).  Why: the register file holds one tile
// (14 values per pixel), so a CU runs load phase -> ten half sweeps -> store phase strictly one after the other — the load
// phase streams its 11 planes at the HBM ceiling (~5 TB/s of unique bytes, 202 us of a 660-us launch at 1080p x 129 pairs,
// profiles/round3/brox/) while the VALU idles, and the sweeps (567 us) leave the memory pipes idle.  Here 7 of the 11 planes
// cross HBM during the sweeps; u, v, du, dv (du / dv are the previous launch's output) go into registers, their loads issued
// BROX_SOR_EARLY_LOADS sweeps before the previous tile's end.  Same per-pixel operations in the same order as k_brox_sor_pk:
// bit-identical.
// Nothing in the tile loop waits for a counter to reach zero while something useful is in flight:
//  * barriers are s_waitcnt lgkmcnt(0) + s_barrier by hand (__syncthreads() would also wait for vmcnt, i.e. for the DMA);
//  * the DMA is the instruction by hand as well (see pf_issue), so that the compiler's own waits stay counted ones;
//  * every wave issues exactly four stores per tile (lanes that own nothing store into a sink), so the tile's top can wait
//    for "all but the four youngest operations": the loads and the DMA, not the acknowledgement of the stores.
// Where a launch's time goes (level 0 at 1080p, 65 pairs, cycle counters of BROX_SOR_DEBUG == 3): sweeps 85 %, taking the
// coefficients out of LDS + the divisions 7 %, top-of-tile wait and barrier 5 %, stores + DMA issue 4 %.  The sweeps run at
// VALU 41 % / LDS 32 % busy — a barrier per half sweep with four waves per SIMD is latency-bound — and the DMA costs them
// about a fifth (the launch without any DMA, wrong flows: 940 us against 1143): LABNOTES.md §11.
#ifndef BROX_SOR_EARLY_LOADS
#define BROX_SOR_EARLY_LOADS 3 // the next tile's u / v / du / dv loads are issued this many sweeps before the tile's end
#endif                         // (measured at 1080p, pairs/s: 0 = after the sweeps 270, 1 274, 2 280, 3 280, 4 278, 5 272)
typedef __attribute__((address_space(3))) void brox_lds_void;
typedef __attribute__((address_space(1))) const void brox_glb_void;
__device__ __forceinline__ void brox_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int S>
__global__ __launch_bounds__(1024) void k_brox_sor_stream(BroxLevelCtx c, int uv_set, int d_src, int n_sweeps, int tiles_x,
                                                          int tiles_per_pair, int total_tiles) {
    constexpr int TW = 64, TH = 64, HALO = 2 * S, NPF = 7;
    enum { F_GX = 0, F_GY, F_IDU, F_IDV, F_ND, F_NU, F_NV };
    // W = (u + du, v + dv) of the tile, the two values of a pixel side by side: a neighbour is ONE ds_read_b64 (2 LDS cycles per
    // wave for both values; two ds_read_b32 are 4).  Columns are split by parity, as in k_brox_sor_pk: the 32 lanes of a lane
    // group read 32 consecutive 8-byte elements = all 64 banks once.
    __shared__ __attribute__((aligned(8))) f2 W[2][TH][TW / 2];
    __shared__ __attribute__((aligned(16))) float PF[NPF][TH][TW]; // the coefficient planes of the tile about to be swept
#define W_AT(ly, lx) W[(lx)&1][ly][(lx) >> 1]
    const int w = c.w, h = c.h, pitch = c.pitch;
    const int pcol = threadIdx.x & 31, prow = threadIdx.x >> 5;
    const int lx0 = 2 * pcol, ly0 = 2 * prow;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float omega = c.omega, om1 = 1.0f - omega;
    constexpr int ROWS_PER_WAVE = 4;
    const int band0 = (ly0 / ROWS_PER_WAVE) * ROWS_PER_WAVE;
    const int xm = max(lx0 - 1, 0), xp = min(lx0 + 2, TW - 1), ym = max(ly0 - 1, 0), yp = min(ly0 + 2, TH - 1);

    auto tile_xy = [&](int k, int &tx, int &ty) { // (row by row.  Block by block — 8 x 4, 4 x 8, 6 x 6 tiles per XCD at a time, so
        ty = k / tiles_x;                         // that halos are shared in L2 on both axes — measured the same to 0.5 %)
        tx = k - ty * tiles_x;
    };
    // tile number of this workgroup's i-th tile: rounds of gridDim.x tiles; inside a round the workgroups of one XCD (ids
    // k, k + 8, ...: dfx_device.h) take one contiguous run, so the halos they share stay in one L2
    const int G = (int)gridDim.x, per_xcd = G >> 3;
    auto tile_of = [&](int i) -> int {
        const int wg = (int)blockIdx.x;
        return (G & 7) ? i * G + wg : i * G + (wg & 7) * per_xcd + (wg >> 3);
    };
    // LDS-DMA of the coefficient planes of a tile (BROX_PL_GX + q, q = 0..6: consecutive planes of the pair slot), 16 bytes per
    // lane: one instruction copies four 64-float tile rows of a plane (lane = row l / 16, columns 4 (l % 16) .. + 3; the LDS
    // side is wave-uniform base + 16 * lane = those rows back to back), wave `wave` takes rows 4 * wave .. + 3 of every
    // plane: 7 instructions per wave and tile.  Rows outside the image read a clamped row; columns outside it are read where
    // they would be (the row's padding, the neighbouring row, for row 0 the tail of the plane in front: all inside the pair
    // slot, whose planes are sized for level 0 and start with u, v) — every such entry is masked when the values are taken
    // out of LDS.  The source address is 8-byte aligned (tile origins are even).
    static_assert(BROX_PL_GY == BROX_PL_GX + 1 && BROX_PL_NV == BROX_PL_GX + 6 && BROX_PL_GX > 0, "see above");
    const float *pf_src = nullptr; // this lane's source in plane GX of the tile being fetched
    auto pf_issue = [&](int t) {
        const int b = t / tiles_per_pair, tile = t - b * tiles_per_pair;
        int tx, ty;
        tile_xy(tile, tx, ty);
        const int x0 = tx * (TW - 2 * HALO) - HALO, y0 = ty * (TH - 2 * HALO) - HALO;
        const int r = 4 * wave + (lane >> 4);
        pf_src = bplane(c, b, BROX_PL_GX) + ((long long)min(max(y0 + r, 0), h - 1) * pitch + (x0 + 4 * (lane & 15)));
        // (the instruction by hand, not __builtin_amdgcn_global_load_lds: the compiler files that builtin as a FLAT access that
        // may touch LDS, after which every vector-memory wait it inserts is vmcnt(0) until one has happened — and the tile
        // loop's point is to never wait for its own stores.  The waits for the DMA are by hand anyway: top of the tile loop.)
        const unsigned lds0 = (unsigned)(size_t)(brox_lds_void *)&PF[0][4 * wave][0];
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const float *g = pf_src + (long long)q * c.plane_stride;
            const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + q * (unsigned)sizeof(PF[0]));
            // M0 = the wave's LDS base; the s_nop is the wait state an LDS-DMA needs after a scalar write of M0 (the hazard
            // recognizer does not look into inline assembly)
            asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "{m0}"(l) : "memory");
        }
    };

    // u, v, du, dv of this thread's patch in tile t (two 8-byte loads per plane; rows / columns outside the image read element
    // 0 and are masked when used): issued for tile i + 1 BROX_SOR_EARLY_LOADS sweeps before tile i's last one — 16 more live
    // registers through those sweeps (128 are in use either way), and the loads' latency is off the tile's critical path
    auto ld2 = [](const float *P, long long off) -> f2 { return *reinterpret_cast<const f2_a8 *>(P + off); };
    f2 n_u[2], n_v[2], n_du[2], n_dv[2];
    auto direct_loads = [&](int tt) { // (branch-free on purpose: see the stores at the end of the tile loop)
        const int b = tt / tiles_per_pair, tile = tt - b * tiles_per_pair;
        int tx, ty;
        tile_xy(tile, tx, ty);
        const int x = tx * (TW - 2 * HALO) - HALO + lx0, y = ty * (TH - 2 * HALO) - HALO + ly0;
        const float *u = bplane(c, b, BROX_PL_U0 + 2 * uv_set), *v = bplane(c, b, BROX_PL_V0 + 2 * uv_set);
        const float *DU = bplane(c, b, du_plane(d_src)), *DV = bplane(c, b, dv_plane(d_src));
        const bool in_x = x >= 0 && x < w;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int o = (in_x && y + k >= 0 && y + k < h) ? (y + k) * pitch + x : 0; // (a level's plane: < 2^31 elements)
            n_u[k] = ld2(u, o);
            n_v[k] = ld2(v, o);
            n_du[k] = ld2(DU, o);
            n_dv[k] = ld2(DV, o);
        }
    };
    float *sink = c.sor_sink + 16 * (int)blockIdx.x;
    int i = 0, t = tile_of(0);
    if (t < total_tiles) {
        pf_issue(t);
        direct_loads(t);
#pragma unroll
        for (int k = 0; k < 4; ++k) // (the four stores a tile's loads are always followed by; apart, so that they stay four)
            *reinterpret_cast<f2_a8 *>(sink + 4 * k) = pk_set(0.0f, 0.0f);
    }
#if BROX_SOR_DEBUG == 3 // measurement build only: where workgroup 0's (and 77's) time goes, printed by the level-0 launches
    long long tk[6] = {0, 0, 0, 0, 0, 0}, t0 = clock64(), t1;
#define BROX_TICK(k) (t1 = clock64(), tk[k] += t1 - t0, t0 = t1)
#else
#define BROX_TICK(k)
#endif
    for (; t < total_tiles; t = tile_of(++i)) {
        const int b = t / tiles_per_pair, tile = t - b * tiles_per_pair;
        int tx, ty;
        tile_xy(tile, tx, ty);
        const int x0 = tx * (TW - 2 * HALO) - HALO; // even
        const int y0 = ty * (TH - 2 * HALO) - HALO; // even
        const int x = x0 + lx0, y = y0 + ly0;       // pixel (row 0, column 0) of the patch
        float *DUo = bplane(c, b, du_plane(d_src ^ 1)), *DVo = bplane(c, b, dv_plane(d_src ^ 1));
        const bool inx0 = x >= 0 && x < w, inx1 = x + 1 >= 0 && x + 1 < w; // x is even: x < 0 puts both columns outside
        const bool iny[3] = {y >= 0 && y < h, y + 1 >= 0 && y + 1 < h, y + 2 >= 0 && y + 2 < h};
        auto mask = [&](f2 r, int k) -> f2 { return pk_set(inx0 && iny[k] ? r.x : 0.0f, inx1 && iny[k] ? r.y : 0.0f); };
        // ---- what is not prefetched: u, v, du, dv of the patch, loaded since the end of the previous tile's sweeps
        f2 r_u[2] = {n_u[0], n_u[1]}, r_v[2] = {n_v[0], n_v[1]}, r_du[2] = {n_du[0], n_du[1]}, r_dv[2] = {n_dv[0], n_dv[1]};
        // ---- the coefficient planes have landed (this wave's DMA: all but the four youngest operations, which are the
        // previous tile's stores; every wave's: the barrier)
        BROX_TICK(0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        BROX_TICK(1);
        brox_lds_barrier();
        BROX_TICK(2);
        auto pf2 = [&](int q, int k) -> f2 { return *reinterpret_cast<const f2_a8 *>(&PF[q][ly0 + k][lx0]); };
        f2 r_gl[2], r_gd[2], r_idu[2], r_idv[2], r_nd[2], r_nu[2], r_nv[2];
        float gxr[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            r_gl[k] = mask(pf2(F_GX, k), k);
            r_gd[k] = mask(pf2(F_GY, k), k);
            r_idu[k] = pf2(F_IDU, k);
            r_idv[k] = pf2(F_IDV, k);
            r_nd[k] = mask(pf2(F_ND, k), k);
            r_nu[k] = mask(pf2(F_NU, k), k);
            r_nv[k] = mask(pf2(F_NV, k), k);
            // GX at x + 2.  In the tile's last patch column that is the next tile's pixel: the value only enters the update
            // of a halo pixel, which nothing reads (k_brox_sor_pk loads it; any value gives the same owned region)
            gxr[k] = PF[F_GX][ly0 + k][min(lx0 + 2, TW - 1)];
            r_u[k] = mask(r_u[k], k);
            r_v[k] = mask(r_v[k], k);
            r_du[k] = mask(r_du[k], k);
            r_dv[k] = mask(r_dv[k], k);
        }
        const f2 r_gy2 = *reinterpret_cast<const f2_a8 *>(&PF[F_GY][min(ly0 + 2, TH - 1)][lx0]); // (bottom patch row: as gxr)
        f2 r_gr[2], r_gu[2], r_gs[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            r_gr[k] = pk_set(inx0 && iny[k] ? r_gl[k].y : 0.0f, (inx1 && iny[k] && x + 2 < w) ? gxr[k] : 0.0f);
            if (k == 0)
                r_gu[k] = pk_set(inx0 && iny[0] ? r_gd[1].x : 0.0f, inx1 && iny[0] ? r_gd[1].y : 0.0f);
            else
                r_gu[k] = pk_set(inx0 && iny[1] && iny[2] ? r_gy2.x : 0.0f, inx1 && iny[1] && iny[2] ? r_gy2.y : 0.0f);
            r_gs[k] = ((r_gl[k] + r_gr[k]) + r_gd[k]) + r_gu[k];
            const f2 den_u = r_idu[k] + r_gs[k], den_v = r_idv[k] + r_gs[k];
            r_idu[k] = mask(pk_div_with_rcp((f2)(1.0f), den_u, pk_refined_rcp(den_u)), k);
            r_idv[k] = mask(pk_div_with_rcp((f2)(1.0f), den_v, pk_refined_rcp(den_v)), k);
            const f2 wu = r_u[k] + r_du[k], wv = r_v[k] + r_dv[k];
            W[0][ly0 + k][pcol] = pk_set(wu.x, wv.x), W[1][ly0 + k][pcol] = pk_set(wu.y, wv.y);
        }
#define BROX_PAIR(r) {pk_set(r[0].x, r[1].y), pk_set(r[0].y, r[1].x)}
        f2 gl[2] = BROX_PAIR(r_gl), gr[2] = BROX_PAIR(r_gr), gd[2] = BROX_PAIR(r_gd), gu[2] = BROX_PAIR(r_gu);
        const f2 gs[2] = BROX_PAIR(r_gs), idu[2] = BROX_PAIR(r_idu), idv[2] = BROX_PAIR(r_idv), nd[2] = BROX_PAIR(r_nd);
        const f2 nu[2] = BROX_PAIR(r_nu), nv[2] = BROX_PAIR(r_nv), uu[2] = BROX_PAIR(r_u), vv[2] = BROX_PAIR(r_v);
        f2 du[2] = BROX_PAIR(r_du), dv[2] = BROX_PAIR(r_dv);
#undef BROX_PAIR
        brox_lds_barrier(); // W complete; every thread has taken its coefficients out of PF
        BROX_TICK(3);
        // The previous tile's stores have had the whole consumption phase to be acknowledged; this wait (a builtin, so that
        // the compiler's counter model sees it) makes it official, and the compiler has no reason left to wait for them —
        // its counter would also count the DMA issued next, which it does not know about — anywhere in the sweeps.
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt and lgkmcnt left alone
#if BROX_SOR_DEBUG != 6 // (6: measurement build only — no LDS-DMA after the first tile, WRONG flows)
        if (tile_of(i + 1) < total_tiles)
            pf_issue(tile_of(i + 1)); // its coefficient planes cross HBM during this tile's sweeps
#endif
#if BROX_SOR_DEBUG == 1 // measurement build only (scripts/build_variant.sh): everything but the sweeps (WRONG flows)
        n_sweeps = 0;
#endif
        // The sweeps, in two runs with the next tile's direct loads between them.  (One loop with the loads under
        // `sw == n_sweeps - BROX_SOR_EARLY_LOADS` makes the compiler assume they may be issued in several iterations: it then
        // waits, vmcnt(3) in the middle of the sweeps, before it overwrites their destination registers, and that counter
        // also counts the DMA in flight.  Measured: 1151 -> 1143 us per launch, i.e. the DMA had landed by then anyway.)
        const int sw_split = max(n_sweeps - BROX_SOR_EARLY_LOADS, 0);
        auto sweep = [&](int sw) {
            const int m = 2 * (n_sweeps - 1 - sw) + 1;
            const bool live = band0 + ROWS_PER_WAVE - 1 >= HALO - m && band0 < TH - HALO + m;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#if BROX_SOR_DEBUG == 4 // measurement build only: the barriers of the sweeps without their bodies (WRONG flows)
                if (false) {
#else
                if (live) {
#endif
                    const int ax = lx0 + q, bx = lx0 + 1 - q;
                    const int axl = q ? lx0 : xm, axr = q ? xp : lx0 + 1;
                    const int bxl = q ? xm : lx0, bxr = q ? lx0 + 1 : xp;
                    // pixel A0 = (ly0, ax), pixel A1 = (ly0 + 1, bx).  The neighbour sums are formed per pixel on (u, v) pairs —
                    // the weights are the same for both — and regrouped into (A0, A1) pairs for the coupled update; every
                    // value goes through the same operations in the same order as in k_brox_sor_pk.
                    const f2 L0 = W_AT(ly0, axl), R0 = W_AT(ly0, axr), D0 = W_AT(ym, ax), U0 = W_AT(ly0 + 1, ax);
                    const f2 L1 = W_AT(ly0 + 1, bxl), R1 = W_AT(ly0 + 1, bxr), D1 = W_AT(ly0, bx), U1 = W_AT(yp, bx);
                    // (the weights of a pixel enter as one half of their (A0, A1) register pair, selected by the multiply's
                    // op_sel.  The empty asm hides from the optimiser that the pairs are loop invariants: it would hoist the
                    // splat out of the sweep loop as a register pair of its own — 16 more registers, and spills.)
                    asm volatile("" : "+v"(gl[q]), "+v"(gr[q]), "+v"(gd[q]), "+v"(gu[q]));
#define BROX_SPLAT(r, c) pk_set(r[q].c, r[q].c)
                    const f2 s0 = (((BROX_SPLAT(gl, x) * L0 + BROX_SPLAT(gr, x) * R0) + BROX_SPLAT(gd, x) * D0) + BROX_SPLAT(gu, x) * U0) -
                                  BROX_SPLAT(gs, x) * pk_set(uu[q].x, vv[q].x);
                    const f2 s1 = (((BROX_SPLAT(gl, y) * L1 + BROX_SPLAT(gr, y) * R1) + BROX_SPLAT(gd, y) * D1) + BROX_SPLAT(gu, y) * U1) -
                                  BROX_SPLAT(gs, y) * pk_set(uu[q].y, vv[q].y);
#undef BROX_SPLAT
                    const f2 t0 = s0 - pk_set(nu[q].x, nv[q].x), t1 = s1 - pk_set(nu[q].y, nv[q].y); // (su - nu, sv - nv)
                    const f2 tu = pk_set(t0.x, t1.x), tv = pk_set(t0.y, t1.y);
                    const f2 du_n = om1 * du[q] + omega * (idu[q] * (tu - nd[q] * dv[q]));
                    const f2 dv_n = om1 * dv[q] + omega * (idv[q] * (tv - nd[q] * du_n));
                    du[q] = du_n;
                    dv[q] = dv_n;
                    const f2 wu = uu[q] + du_n, wv = vv[q] + dv_n;
                    W_AT(ly0, ax) = pk_set(wu.x, wv.x);
                    W_AT(ly0 + 1, bx) = pk_set(wu.y, wv.y);
                }
                brox_lds_barrier();
            }
        };
        for (int sw = 0; sw < sw_split; ++sw)
            sweep(sw);
        direct_loads(min(tile_of(i + 1), total_tiles - 1)); // (after the last tile: a reload nobody uses — the shape is fixed)
        for (int sw = sw_split; sw < n_sweeps; ++sw)
            sweep(sw);
        BROX_TICK(4);
        // ---- store the owned region into the other du / dv set.  EVERY lane stores, exactly four instructions per wave: a
        // lane that owns nothing stores into the workgroup's sink.  That fixed count is what lets the next tile wait for its
        // loads with vmcnt(4) — the counter retires in issue order, so vmcnt(0) would also wait for these stores to be
        // acknowledged by L2, once per tile (a workgroup per tile never waits for its stores: its waves just end).
        // With an odd width the last column's 8-byte store puts its second half into the row's padding (pitch is w rounded up
        // to 64), which every reader masks.
        {
            const bool own_x = lx0 >= HALO && lx0 < TW - HALO && x < w;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ly = ly0 + k, yy = y + k;
                const bool own = own_x && ly >= HALO && ly < TH - HALO && yy < h;
                const long long oo = (long long)yy * pitch + x;
                const f2 a = k == 0 ? pk_set(du[0].x, du[1].x) : pk_set(du[1].y, du[0].y);
                const f2 bq = k == 0 ? pk_set(dv[0].x, dv[1].x) : pk_set(dv[1].y, dv[0].y);
                *reinterpret_cast<f2_a8 *>(own ? DUo + oo : sink) = a;
                *reinterpret_cast<f2_a8 *>(own ? DVo + oo : sink + 2) = bq;
            }
        }
        // (the next tile's first barrier follows its own loads: W of this tile is no longer read by then — every wave
        // has passed the last sweep barrier)
    }
#if BROX_SOR_DEBUG == 3
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 77) && total_tiles > 50000)
        printf("sor_stream wg %d tiles %d cycles: stores+issue %lld wait_vm %lld barrier %lld consume %lld sweeps %lld\n",
               (int)blockIdx.x, i, tk[0], tk[1], tk[2], tk[3], tk[4]);
#endif
#undef BROX_TICK
#undef W_AT
}

__global__ __launch_bounds__(256) void k_brox_add_increment(BroxLevelCtx c, int uv_set, int d_set) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= c.w || y >= c.h)
        return;
    const int b = blockIdx.z;
    const long long o = (long long)y * c.pitch + x;
    float *u = bplane(c, b, BROX_PL_U0 + 2 * uv_set), *v = bplane(c, b, BROX_PL_V0 + 2 * uv_set);
    u[o] = u[o] + bplane(c, b, du_plane(d_set))[o];
    v[o] = v[o] + bplane(c, b, dv_plane(d_set))[o];
}

// The bicubic weights of a window are separable and the same for u and v: they are formed once per pixel (<= 5 + 5
// evaluations of brox_bicubic_w instead of up to 2 x 30), every tap's weight is still the single product w(x - cx) * w(y - cy)
// and both sums still take their taps rows outer, columns inner: the same bits (round 6; the kernel was 4 % of Brox).
__global__ __launch_bounds__(256) void k_brox_prolongate(BroxLevelCtx c, int uv_set, int dw, int dh, int dpitch,
                                                         float factor, float mul) {
    const DfxBlockXY blk = dfx_block_xy(); // XCD-aware (dfx_device.h): neighbouring rows share one L2
    const int ix = blk.x * 64 + (threadIdx.x & 63);
    const int iy = blk.y * 4 + (threadIdx.x >> 6);
    if (ix >= dw || iy >= dh)
        return;
    const int b = blockIdx.z, sw = c.w, sh = c.h;
    const float y = (float)iy * factor, x = (float)ix * factor;
    const int y0 = max((int)ceilf(y - 2.0f), 0), y1 = min((int)floorf(y + 2.0f), sh - 1);
    const int x0 = max((int)ceilf(x - 2.0f), 0), x1 = min((int)floorf(x + 2.0f), sw - 1);
    float wxs[5]; // [ceil(x - 2), floor(x + 2)] holds at most 5 integers
#pragma unroll
    for (int i = 0; i < 5; ++i)
        wxs[i] = brox_bicubic_w(x - (float)(x0 + i));
    const float *su = bplane(c, b, BROX_PL_U0 + 2 * uv_set), *sv = bplane(c, b, BROX_PL_V0 + 2 * uv_set);
    float sum_u = 0.f, sum_v = 0.f, wsum = 0.f;
    for (int cy = y0; cy <= y1; ++cy) {
        const float wy = brox_bicubic_w(y - (float)cy);
        const long long ro = (long long)cy * c.pitch + x0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (x0 + i <= x1) {
                const float wgt = wxs[i] * wy;
                sum_u = sum_u + wgt * su[ro + i];
                sum_v = sum_v + wgt * sv[ro + i];
                wsum = wsum + wgt;
            }
        }
    }
    const long long o = (long long)iy * dpitch + ix;
    bplane(c, b, BROX_PL_U0 + 2 * (uv_set ^ 1))[o] = ((wsum == 0.0f) ? 0.0f : sum_u / wsum) * mul;
    bplane(c, b, BROX_PL_V0 + 2 * (uv_set ^ 1))[o] = ((wsum == 0.0f) ? 0.0f : sum_v / wsum) * mul;
}

__global__ __launch_bounds__(256) void k_brox_merge(BroxLevelCtx c, int uv_set, float *out, long long out_stride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= c.w || y >= c.h)
        return;
    const int b = blockIdx.z;
    const long long o = (long long)y * c.pitch + x;
    float2 v;
    v.x = bplane(c, b, BROX_PL_U0 + 2 * uv_set)[o];
    v.y = bplane(c, b, BROX_PL_V0 + 2 * uv_set)[o];
    reinterpret_cast<float2 *>(out + (long long)b * out_stride)[(long long)y * c.w + x] = v;
}

// ------------------------------------------------------------------------------------------------ launchers

void brox_launch_u8_to_f32(hipStream_t s, const unsigned char *src, long long src_frame_stride, long long src_pitch,
                           const int *frame_slots, int n_frames, float *frames, long long frame_stride, int w, int h,
                           int pitch, float scale) {
    hipLaunchKernelGGL(k_brox_u8_to_f32, bgrid(w, h, n_frames), dim3(256), 0, s, src, src_frame_stride, src_pitch,
                       frame_slots, frames, frame_stride, w, h, pitch, scale);
}
void brox_launch_downsample(hipStream_t s, float *frames, long long frame_stride, const int *frame_slots,
                            int n_frames, long long src_off, int sw, int sh, int spitch, long long dst_off, int dw,
                            int dh, int dpitch, float factor) {
    hipLaunchKernelGGL(k_brox_downsample, bgrid(dw, dh, n_frames), dim3(256), 0, s, frames, frame_stride, frame_slots,
                       src_off, sw, sh, spitch, dst_off, dw, dh, dpitch, factor);
}
void brox_launch_deriv(hipStream_t s, float *frames, long long frame_stride, const int *frame_slots, int n_frames,
                       long long src_off, long long dst_off, int w, int h, int pitch, int axis) {
    hipLaunchKernelGGL(k_brox_deriv, bgrid(w, h, n_frames), dim3(256), 0, s, frames, frame_stride, frame_slots, src_off,
                       dst_off, w, h, pitch, axis);
}
void brox_launch_level_init(hipStream_t s, const BroxLevelCtx &c, int uv_set, int zero_uv) {
    hipLaunchKernelGGL(k_brox_level_init, bgrid(c.w, c.h, c.n_pairs), dim3(256), 0, s, c, uv_set, zero_uv);
}
void brox_launch_stage1(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_set) {
    hipLaunchKernelGGL(k_brox_stage1, bgrid(c.w, c.h, c.n_pairs), dim3(256), 0, s, c, uv_set, d_set);
}
void brox_launch_stage2(hipStream_t s, const BroxLevelCtx &c) {
    hipLaunchKernelGGL(k_brox_stage2, bgrid(c.w, c.h, c.n_pairs), dim3(256), 0, s, c);
}
void brox_launch_sor(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_set, int color) {
    hipLaunchKernelGGL(k_brox_sor, bgrid((c.w + 1) / 2, c.h, c.n_pairs), dim3(256), 0, s, c, uv_set, d_set, color);
}
// 64 x 64 tile, 5 sweeps per launch: the reference's 10 solver iterations are two launches (measured: 64x64x5 192
// pairs/s at 1080p against 153 / 141 for 3 / 2 sweeps and 145 for 64x32 and 128x32 tiles).
constexpr int BROX_SWEEPS = 5;
int brox_fused_sweeps() { return BROX_SWEEPS; }
void brox_launch_sor_fused(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_src, int n_sweeps) {
    constexpr int TW = 64, TH = 64, S = BROX_SWEEPS;
    const int tiles_x = (c.w + (TW - 4 * S) - 1) / (TW - 4 * S), tiles_y = (c.h + (TH - 4 * S) - 1) / (TH - 4 * S);
    const dim3 grid(tiles_x * tiles_y, 1, c.n_pairs), block(TW * TH / 4);
    const int total = tiles_x * tiles_y * c.n_pairs;
    // one persistent workgroup per CU (a multiple of 8: the XCD-aware tile order needs whole rounds) where every CU gets at
    // least four tiles; below that the static tile order leaves CUs idle that one workgroup per tile would have fed
    if (c.sor_stream && total >= 4 * c.sor_stream) {
        const int wgs = std::min(std::max(c.sor_stream / 8, 1) * 8, total);
        hipLaunchKernelGGL((k_brox_sor_stream<S>), dim3(wgs), block, 0, s, c, uv_set, d_src, n_sweeps, tiles_x,
                           tiles_x * tiles_y, total);
    } else if (c.sor_progress)
        hipLaunchKernelGGL((k_brox_sor_pk<S, 1>), grid, block, 0, s, c, uv_set, d_src, n_sweeps, tiles_x);
    else
        hipLaunchKernelGGL((k_brox_sor_pk<S, 0>), grid, block, 0, s, c, uv_set, d_src, n_sweeps, tiles_x);
}
void brox_launch_add_increment(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_set) {
    hipLaunchKernelGGL(k_brox_add_increment, bgrid(c.w, c.h, c.n_pairs), dim3(256), 0, s, c, uv_set, d_set);
}
void brox_launch_prolongate(hipStream_t s, const BroxLevelCtx &c, int uv_set, int dw, int dh, int dpitch, float factor,
                            float mul) {
    hipLaunchKernelGGL(k_brox_prolongate, bgrid(dw, dh, c.n_pairs), dim3(256), 0, s, c, uv_set, dw, dh, dpitch, factor,
                       mul);
}
void brox_launch_merge(hipStream_t s, const BroxLevelCtx &c, int uv_set, float *out, long long out_stride) {
    hipLaunchKernelGGL(k_brox_merge, bgrid(c.w, c.h, c.n_pairs), dim3(256), 0, s, c, uv_set, out, out_stride);
}
