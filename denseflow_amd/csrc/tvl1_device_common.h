// tvl1_device_common.h — device helpers shared by tvl1_kernels.hip (step kernels, level control) and tvl1_warp_kernels.hip
// (the backward warp as its own kernel): plane addressing, the deterministic reductions, the arrival ticket that lets the
// last workgroup of a pair advance its state machine, and A.5's backward warp of one pixel.  Device code only.
#pragma once

#include <hip/hip_runtime.h>

#include "dfx_device.h"
#include "tvl1_math.h"

#define AGENT __HIP_MEMORY_SCOPE_AGENT

// ------------------------------------------------------------------------------------------------
// small helpers

__device__ __forceinline__ float *pair_plane(const Tvl1LevelCtx &c, int pair, int plane) {
    return c.planes + (long long)pair * c.slot_stride + (long long)plane * c.plane_stride;
}

// Buffer addressing for the per-row accesses of the tile kernels (round 6).  A thread reads and writes its 8 tile rows in up to
// 9 planes of its pair slot: through plain pointers every one of those accesses pays a 64-bit shift-and-add on the vector ALU
// (plane base + 4 * element offset; the compiler's 1311 of the step kernel's 7612 static vector instructions, 2121 of the
// warp-and-head kernel's 15 719) in kernels that are bound by vector-ALU issue.  A buffer instruction takes the slot's base from
// a 128-bit descriptor in scalar registers, the plane's byte offset from ONE scalar register (soffset) and the pixel's byte
// offset inside the plane from ONE 32-bit vector register: no vector instruction per access.  Offsets are bytes in 32 bits: the
// engine refuses frame sizes whose pair slot reaches 4 GB (16 planes: beyond 8192 x 8192).  The descriptor's size field bounds
// vector + scalar offset (measured on gfx950, scripts/round6/probe_buffer.py: the scalar offset IS part of the range check), so it is
// set to the slot's size: every plane of the slot is in range, anything beyond it reads 0 / is not written.
// Only dword accesses: this compiler's __builtin_amdgcn_raw_buffer_load_b128 emits a one-dword load and splats it.
typedef __amdgpu_buffer_rsrc_t dfx_rsrc;
__device__ __forceinline__ dfx_rsrc dfx_make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000); // raw buffer, 32-bit data
}
__device__ __forceinline__ dfx_rsrc pair_rsrc(const Tvl1LevelCtx &c, int pair) { // the pair's slot: plane q at soffset q * plane_bytes
    return dfx_make_rsrc(c.planes + (long long)pair * c.slot_stride, (unsigned)(c.slot_stride * 4));
}
__device__ __forceinline__ unsigned plane_soff(const Tvl1LevelCtx &c, int plane) { return (unsigned)plane * (unsigned)(c.plane_stride * 4); }
__device__ __forceinline__ float buf_ld(dfx_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_st(dfx_rsrc r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

// The level context of a kernel whose FIRST parameter is a Tvl1LevelCtx by value, read from the kernel-argument segment at the
// point of use — for the ONCE-PER-LEVEL paths (finish_level, the in-kernel warp phase of the non-default configurations).  A
// by-value parameter is loaded into scalar registers once, at the top; the fields only those paths need then live across the
// whole kernel, and the step kernel, short of scalar registers, parks them in VGPR lanes (v_writelane / v_readlane: vector-ALU
// instructions, in a kernel bound by vector-ALU issue).  Behind this reference such a path loads them (s_load, scalar cache)
// when it runs; the empty asm makes the pointer opaque, so the compiler cannot turn the loads back into a use of the parameter.
// Measured (scripts/round6/r6_gpu39.sh): step kernel 105 -> 86 lane writes, 250 -> 152 lane reads (static), 501.4 vs 496.4 pairs/s.
// NOT for the hot phases: the same trick in front of the store phase and the epilogue is 2 % slower (LABNOTES section 11) — the
// scalar loads then sit on every wave's path, and in the warp-and-head kernel it made the spilling worse.
__device__ __forceinline__ const Tvl1LevelCtx &dfx_kernarg_ctx() {
    auto p = (const __attribute__((address_space(4))) Tvl1LevelCtx *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const Tvl1LevelCtx *)p;
}

__device__ __forceinline__ double wave_reduce_sum_f64(double v) {
    // fixed butterfly order -> deterministic
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

// Deterministic workgroup sum (any whole number of waves up to 8): wave butterflies, then the waves in
// index order.  Result valid in thread 0.
__device__ __forceinline__ double block_reduce_sum_f64(double v, double *lds8) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_reduce_sum_f64(v);
    __syncthreads(); // lds8 may still be read from a previous use
    if (lane == 0)
        lds8[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        r = lds8[0];
        for (int i = 1; i < nw; ++i)
            r = r + lds8[i];
    }
    return r;
}

// Arrival ticket: true (in every thread) for the last workgroup of this pair to arrive.
// A workgroup's partial sum is published beforehand with a write-through agent-scope 8-byte atomic
// store and drained, then the ticket is taken; the last arriver reads the partials with agent-scope
// 8-byte atomic loads (cdna_hip_programming.md Guideline 16, "8-B agent atomics both sides").
// Nothing here depends on dispatch order or on which XCD a workgroup runs.
__device__ __forceinline__ bool arrive_is_last(Tvl1State *st, unsigned nblk, int *lds_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, AGENT);
        *lds_flag = (old == nblk - 1u) ? 1 : 0;
    }
    __syncthreads();
    return *lds_flag != 0;
}

__device__ __forceinline__ void publish_partial(double *slot, double v) {
    __hip_atomic_store((unsigned long long *)slot, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       AGENT);
}
__device__ __forceinline__ double read_partial(const double *slot) {
    const unsigned long long bits =
        __hip_atomic_load((const unsigned long long *)slot, __ATOMIC_RELAXED, AGENT);
    return __longlong_as_double((long long)bits);
}

// Called by the one thread that moved a pair to TVL1_PH_LEVEL_DONE.
__device__ __forceinline__ void finish_level(const Tvl1LevelCtx &c, int pair, const Tvl1State &s, int step_id) {
    int *io = c.iters_out + ((long long)pair * DFX_LVL_MAX + c.level) * TVL1_MAX_WARPS;
    for (int i = 0; i < TVL1_MAX_WARPS; ++i)
        io[i] = (i < c.loop.warps) ? s.iters[i] : 0;
    c.checks_out[((long long)pair * DFX_LVL_MAX + c.level) * 2 + 0] = s.n_checks;
    c.checks_out[((long long)pair * DFX_LVL_MAX + c.level) * 2 + 1] = step_id + 1; // steps that did work
    c.work_out[((long long)pair * DFX_LVL_MAX + c.level) * 2 + 0] = s.step_work;
    c.work_out[((long long)pair * DFX_LVL_MAX + c.level) * 2 + 1] = s.head_work;
    const unsigned old = __hip_atomic_fetch_add(c.level_done_count, 1u, __ATOMIC_RELAXED, AGENT);
    if (old == (unsigned)c.n_pairs - 1u) {
        __hip_atomic_store(c.level_done_count, 0u, __ATOMIC_RELAXED, AGENT);
        __hip_atomic_store((int *)c.host_done_flag, c.done_token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// A.5 backward warp of one pixel

struct WarpOut {
    float I1wx, I1wy, grad, rho_c;
};

// Upstream sums w(cx)*w(cy)*tex(cx,cy) over cx in [ceil(wx-2), floor(wx+2)], cy likewise, rows outer,
// columns inner.  That window is 4 taps wide, or 5 when the coordinate is an exact integer - and in
// every case the tap at distance >= 2 has weight exactly 0 (bicubicCoeff(2) == 0), so it adds +-0 to
// each sum.  Evaluating taps ceil(w-2) .. ceil(w-2)+3 only, in the same order, with the 1-D weights
// hoisted out of the 2-D loop (the product is the same single multiply), gives the same bits.
// Non-finite flow values (upstream does not guard them either) just produce non-finite output here.
// 16-byte load with 4-byte alignment: the four taps of one window row are consecutive floats at an
// arbitrary pixel offset (gfx950 global loads handle unaligned dwordx4).  One such load replaces four
// dword gathers — the address unit spends ~16 cycles per 64-lane load instruction whatever its width,
// so the instruction count, not the byte count, bounds this kernel.
typedef float float4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct WarpTaps { // the 4x4 source window of one pixel, three planes
    float t1[4][4], tx[4][4], ty[4][4];
};

// Issue the loads of one pixel's window (no arithmetic on the loaded values: several pixels' loads can
// be in flight before the first is consumed).
__device__ __forceinline__ void warp_fetch(WarpTaps &T, const float *I1, const float *I1x, const float *I1y, int w,
                                           int h, int pitch, int x, int y, float u1v, float u2v) {
    const float fx0 = ceilf(((float)x + u1v) - 2.0f), fy0 = ceilf(((float)y + u2v) - 2.0f);
    // clamp before the int conversion so NaN/Inf cannot index out of range
    const int xmin = (int)fminf(fmaxf(fx0, -4.0f), (float)w + 4.0f);
    const int ymin = (int)fminf(fmaxf(fy0, -4.0f), (float)h + 4.0f);
    long long ro[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        ro[j] = (long long)min(max(ymin + j, 0), h - 1) * pitch; // clamp-to-edge point sampling (rows)
    if (xmin >= 0 && xmin + 3 <= w - 1) { // the row window is not clipped by the left/right border
#pragma unroll
        for (int jy = 0; jy < 4; ++jy) {
            const long long r = ro[jy] + xmin;
            const float4_a4 a = *reinterpret_cast<const float4_a4 *>(I1 + r);
            const float4_a4 bq = *reinterpret_cast<const float4_a4 *>(I1x + r);
            const float4_a4 cq = *reinterpret_cast<const float4_a4 *>(I1y + r);
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                T.t1[jy][jx] = a[jx];
                T.tx[jy][jx] = bq[jx];
                T.ty[jy][jx] = cq[jx];
            }
        }
    } else {
#pragma unroll
        for (int jy = 0; jy < 4; ++jy)
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const long long r = ro[jy] + min(max(xmin + jx, 0), w - 1);
                T.t1[jy][jx] = I1[r];
                T.tx[jy][jx] = I1x[r];
                T.ty[jy][jx] = I1y[r];
            }
    }
}

__device__ __forceinline__ WarpOut warp_finish(const WarpTaps &T, float I0v, int x, int y, float u1v, float u2v) {
    const float wx = (float)x + u1v;
    const float wy = (float)y + u2v;
    const float fx0 = ceilf(wx - 2.0f), fy0 = ceilf(wy - 2.0f);
    float cwx[4], cwy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cwx[j] = tvl1_bicubic_coeff(wx - (fx0 + (float)j));
        cwy[j] = tvl1_bicubic_coeff(wy - (fy0 + (float)j));
    }
    float sum = 0.0f, sumx = 0.0f, sumy = 0.0f, wsum = 0.0f;
#pragma unroll
    for (int jy = 0; jy < 4; ++jy) {
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const float wgt = cwx[jx] * cwy[jy];
            sum = sum + wgt * T.t1[jy][jx];
            sumx = sumx + wgt * T.tx[jy][jx];
            sumy = sumy + wgt * T.ty[jy][jx];
            wsum = wsum + wgt;
        }
    }
    const float coeff = 1.0f / wsum;
    const float I1w = sum * coeff;
    WarpOut o;
    o.I1wx = sumx * coeff;
    o.I1wy = sumy * coeff;
    o.grad = o.I1wx * o.I1wx + o.I1wy * o.I1wy;
    o.rho_c = ((I1w - o.I1wx * u1v) - o.I1wy * u2v) - I0v;
    return o;
}

__device__ __forceinline__ WarpOut warp_backward_px(const float *I0, const float *I1, const float *I1x,
                                                    const float *I1y, int w, int h, int pitch, int x, int y,
                                                    float u1v, float u2v) {
    WarpTaps T;
    warp_fetch(T, I1, I1x, I1y, w, h, pitch, x, y, u1v, u2v);
    return warp_finish(T, I0[(long long)y * pitch + x], x, y, u1v, u2v);
}

// The same arithmetic with the I0 value passed in (the dedicated warp kernel loads it with the flow).
__device__ __forceinline__ WarpOut warp_backward_px_v(const float *I1, const float *I1x, const float *I1y, int w, int h,
                                                      int pitch, int x, int y, float u1v, float u2v, float I0v) {
    WarpTaps T;
    warp_fetch(T, I1, I1x, I1y, w, h, pitch, x, y, u1v, u2v);
    return warp_finish(T, I0v, x, y, u1v, u2v);
}

