// tvl1_warp_kernels.hip — the backward warp of the -a=tvl1 hot path as kernels of their own (A.5; upstream:
// warpBackwardKernel of cv::cuda::OpticalFlowDual_TVL1, reference call site src/denseflow_gpu.cpp:327).
// One launch in front of every step launch: pairs in phase WARP are warped, the last workgroup of such a pair starts
// its inner loop at this very step (tvl1_ctrl.h), all other pairs cost one state load per workgroup.
// Compiled with -ffp-contract=off (see tvl1_math.h).
#include <hip/hip_runtime.h>

#include "dfx_device.h"
#include "tvl1_device_common.h"
#include "tvl1_kernels.h"

// ------------------------------------------------------------------------------------------------
// The backward warp as its own kernel (the tuned default; `split_warp`).  Inside the step kernel the warp phase runs
// with the step kernel's footprint — 168 VGPRs, 36 KB of LDS: 12 waves per CU — and it is a dependent gather
// (flow -> address -> 4x4 window of three planes -> weights), latency-bound with two thirds of the wave cycles waiting.
// On its own it needs 84 registers and no LDS: 24 waves per CU hide that latency (385 -> 400 pairs/s at 1080p; tighter
// register budgets for 7 / 8 waves per SIMD spill and are slower: 375 / 349).  Every step launches this
// kernel first (pairs that are not in phase WARP cost one state load per workgroup), then the step kernel, which
// finds those pairs in phase ITER with their first segment starting at THIS step: a warp no longer occupies a step
// slot of its own (25 fewer launches per pair at the reference's 5 levels x 5 warps).
// One workgroup = a 64-column x 16-row strip; lane = column, wave w takes rows w, w+4, w+8, w+12.
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_tvl1_warp(Tvl1LevelCtx c, int step_id, int strips_x) {
    __shared__ int lds_flag;
    const int b = blockIdx.z;
    Tvl1State *st = c.state + b;
    if (st->phase != TVL1_PH_WARP)
        return;
    const int strip = dfx_block_linear(); // XCD-aware: the bicubic windows of neighbouring strips overlap
    const int sx = strip % strips_x, sy = strip / strips_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = sx * 64 + lane;
    const int cur = st->cur;
    const PairDesc pd = c.pairs[b];
    const float *I0 = c.frame_I + (long long)pd.frame_a * c.frame_stride + c.lvl_off;
    const long long fb = (long long)pd.frame_b * c.frame_stride + c.lvl_off;
    const float *P1 = c.frame_I + fb, *P1x = c.frame_Ix + fb, *P1y = c.frame_Iy + fb;
    const float *u1p = pair_plane(c, b, PL_U1_0 + 2 * cur), *u2p = pair_plane(c, b, PL_U2_0 + 2 * cur);
    float *o_wx = pair_plane(c, b, PL_I1WX), *o_wy = pair_plane(c, b, PL_I1WY), *o_rc = pair_plane(c, b, PL_RHOC);
    float u1r[4], u2r[4], i0r[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = sy * 16 + wave + 4 * j;
        ok[j] = x < c.w && y < c.h;
        const long long o = ok[j] ? ((long long)y * c.pitch + x) : 0;
        u1r[j] = u1p[o];
        u2r[j] = u2p[o];
        i0r[j] = I0[o];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = sy * 16 + wave + 4 * j;
        if (ok[j]) {
            const WarpOut r = warp_backward_px_v(P1, P1x, P1y, c.w, c.h, c.pitch, x, y, u1r[j], u2r[j], i0r[j]);
            const long long o = (long long)y * c.pitch + x;
            o_wx[o] = r.I1wx;
            o_wy[o] = r.I1wy;
            o_rc[o] = r.rho_c;
        }
    }
    if (arrive_is_last(st, gridDim.x, &lds_flag) && threadIdx.x == 0) {
        // the step kernel of THIS step id follows in the stream: the loop's first segment starts here
        tvl1_begin_loop(*st, c.loop, step_id - 1);
        if (st->phase == TVL1_PH_LEVEL_DONE)
            finish_level(c, b, *st, step_id);
        __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, AGENT);
    }
}

// The backward warp through an LDS tile (round 5; the default — DFX_VAR_TVL1_WARP_GATHER selects k_tvl1_warp above).
// The gather kernel is bound by the texture-address unit: 12 unaligned dwordx4 gathers per pixel, ~16 cycles per 64-lane
// load instruction whatever its width.  Flows are small where most pixels are (|u| of a few pixels at level 0, less at
// the coarse levels), so the 4x4 windows of a 64 x SR strip almost always lie inside the strip grown by MARGIN + 2
// pixels: that region of I1, I1x, I1y is copied into LDS with aligned 16-byte row loads (clamp-to-edge applied while
// copying, so a tile entry IS the point-sampled texture value) — 1.3 load instructions per pixel instead of 12 — and a
// pixel whose window lies inside the tile reads its 48 taps from LDS; any other pixel (|flow| beyond MARGIN: occlusion
// borders, large motion) takes the global path of warp_fetch.  Either way the WarpTaps are the same floats and
// warp_finish is the same code: bit-identical by construction (tests/test_tvl1_gpu.py runs both forms, incl. pairs whose
// flow reaches 51 px).  154 -> 137 us per launch at 1080p x 129 pairs, 449.5 -> 458.4 pairs/s.
template <int MARGIN, int SR, int WPS>
__global__ __launch_bounds__(256, WPS) void k_tvl1_warp_lds(Tvl1LevelCtx c, int step_id, int strips_x) {
    // horizontal halo 8 (rows of the tile start 32-byte aligned in the plane: 16-byte row loads), vertical MARGIN + 2
    constexpr int HX = 8, HY = MARGIN + 2, TWL = 64 + 2 * HX, THL = SR + 2 * HY, RPT = SR / 4; // SR strip rows, RPT per thread
    static_assert(MARGIN + 2 <= HX, "the window must fit the horizontal halo");
    __shared__ __attribute__((aligned(16))) float tile[3][THL][TWL];
    __shared__ int lds_flag;
    const int b = blockIdx.z;
    Tvl1State *st = c.state + b;
    if (st->phase != TVL1_PH_WARP)
        return;
    const int strip = dfx_block_linear();
    const int sx = strip % strips_x, sy = strip / strips_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = sx * 64 + lane;
    const int cur = st->cur;
    const PairDesc pd = c.pairs[b];
    const float *I0 = c.frame_I + (long long)pd.frame_a * c.frame_stride + c.lvl_off;
    const long long fb = (long long)pd.frame_b * c.frame_stride + c.lvl_off;
    const float *P1 = c.frame_I + fb, *P1x = c.frame_Ix + fb, *P1y = c.frame_Iy + fb;
    const float *u1p = pair_plane(c, b, PL_U1_0 + 2 * cur), *u2p = pair_plane(c, b, PL_U2_0 + 2 * cur);
    float *o_wx = pair_plane(c, b, PL_I1WX), *o_wy = pair_plane(c, b, PL_I1WY), *o_rc = pair_plane(c, b, PL_RHOC);
    const int tx0 = sx * 64 - HX, ty0 = sy * SR - HY; // image coordinates of tile[.][0][0]
    float u1r[RPT], u2r[RPT], i0r[RPT];
    bool ok[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int y = sy * SR + wave + 4 * j;
        ok[j] = x < c.w && y < c.h;
        const long long o = ok[j] ? ((long long)y * c.pitch + x) : 0;
        u1r[j] = u1p[o];
        u2r[j] = u2p[o];
        i0r[j] = I0[o];
    }
    // tile copy, 16 bytes per lane: float4 q of tile row r covers image columns tx0 + 4q .. + 3 of row clamp(ty0 + r).
    // Clamp-to-edge is applied here, so a tile entry IS the point-sampled texture value: a float4 left of the image is
    // column 0 four times, one right of it column w - 1, one that straddles the right border is patched per element (the
    // row pitch is a multiple of 64 floats >= w, so the aligned 16-byte load itself never leaves the row).
    constexpr int Q = TWL / 4;
    for (int i = threadIdx.x; i < THL * Q; i += 256) {
        const int r = i / Q, q = i - r * Q;
        const long long ro = (long long)min(max(ty0 + r, 0), c.h - 1) * c.pitch;
        const int gx = tx0 + 4 * q;
        const int lx4 = min(max(gx, 0), ((c.w - 1) >> 2) << 2); // aligned, inside the row
        const float4 a = *reinterpret_cast<const float4 *>(P1 + ro + lx4);
        const float4 bq = *reinterpret_cast<const float4 *>(P1x + ro + lx4);
        const float4 cq = *reinterpret_cast<const float4 *>(P1y + ro + lx4);
        if (gx >= 0 && gx + 3 <= c.w - 1) { // inside the row: three 16-byte LDS stores
            *reinterpret_cast<float4 *>(&tile[0][r][4 * q]) = a;
            *reinterpret_cast<float4 *>(&tile[1][r][4 * q]) = bq;
            *reinterpret_cast<float4 *>(&tile[2][r][4 * q]) = cq;
            continue;
        }
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w}, cv[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = min(max(gx + e, 0), c.w - 1) - lx4; // 0..3: which element of the loaded float4 column gx + e clamps to
            tile[0][r][4 * q + e] = k == 0 ? av[0] : k == 1 ? av[1] : k == 2 ? av[2] : av[3];
            tile[1][r][4 * q + e] = k == 0 ? bv[0] : k == 1 ? bv[1] : k == 2 ? bv[2] : bv[3];
            tile[2][r][4 * q + e] = k == 0 ? cv[0] : k == 1 ? cv[1] : k == 2 ? cv[2] : cv[3];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int y = sy * SR + wave + 4 * j;
        if (ok[j]) {
            const float u1v = u1r[j], u2v = u2r[j];
            // the window's first tap, exactly as warp_fetch derives it
            const float fx0 = ceilf(((float)x + u1v) - 2.0f), fy0 = ceilf(((float)y + u2v) - 2.0f);
            const int xmin = (int)fminf(fmaxf(fx0, -4.0f), (float)c.w + 4.0f);
            const int ymin = (int)fminf(fmaxf(fy0, -4.0f), (float)c.h + 4.0f);
            const int lx = xmin - tx0, ly = ymin - ty0;
            WarpTaps T;
            if (lx >= 0 && lx + 3 < TWL && ly >= 0 && ly + 3 < THL) {
#pragma unroll
                for (int jy = 0; jy < 4; ++jy)
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        T.t1[jy][jx] = tile[0][ly + jy][lx + jx];
                        T.tx[jy][jx] = tile[1][ly + jy][lx + jx];
                        T.ty[jy][jx] = tile[2][ly + jy][lx + jx];
                    }
            } else {
                warp_fetch(T, P1, P1x, P1y, c.w, c.h, c.pitch, x, y, u1v, u2v);
            }
            const WarpOut r = warp_finish(T, i0r[j], x, y, u1v, u2v);
            const long long o = (long long)y * c.pitch + x;
            o_wx[o] = r.I1wx;
            o_wy[o] = r.I1wy;
            o_rc[o] = r.rho_c;
        }
    }
    if (arrive_is_last(st, gridDim.x, &lds_flag) && threadIdx.x == 0) {
        tvl1_begin_loop(*st, c.loop, step_id - 1);
        if (st->phase == TVL1_PH_LEVEL_DONE)
            finish_level(c, b, *st, step_id);
        __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, AGENT);
    }
}

void tvl1_launch_warp(hipStream_t s, const Tvl1LevelCtx &c, int step_id) {
    const int strips_x = (c.w + 63) / 64;
    const int strips_y = (c.h + 15) / 16; // both forms: 64 x 16 strips (32-row strips and 5 waves per SIMD measured: no better)
    const dim3 grid(strips_x * strips_y, 1, c.n_pairs);
    if (c.warp_lds)
        hipLaunchKernelGGL((k_tvl1_warp_lds<4, 16, 4>), grid, dim3(256), 0, s, c, step_id, strips_x);
    else
        hipLaunchKernelGGL(k_tvl1_warp<5>, grid, dim3(256), 0, s, c, step_id, strips_x);
}
