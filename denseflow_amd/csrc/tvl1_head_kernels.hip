// tvl1_head_kernels.hip — the backward warp of the -a=tvl1 hot path fused with the head of the inner loop it starts
// (round 6; upstream: warpBackwardKernel + the first two estimateUKernel / estimateDualVariablesKernel rounds of
// cv::cuda::OpticalFlowDual_TVL1, reference call site src/denseflow_gpu.cpp:327).
//
// Why: on converging content a pair spends 25 warps per pyramid, and 20 of them (warps 1-4 of every level) consist of the
// warp and ONE two-iteration step — the first convergence check is due at iteration 1 and passes.  As two launches that
// costs, per pixel, the warp's 3 + 3 plane reads and 3 plane writes plus the step's 9 reads (x 1.5 with its 4-pixel halo)
// and 6 writes, in kernels that run at 4.4-4.9 TB/s of their own bytes.  Here one workgroup
//   1. copies the (64 + 16) x (32 + 12) neighbourhood of I1, I1x, I1y of its 64 x 32 tile into LDS (aligned 16-byte row
//      loads, clamp-to-edge applied while copying — the tile of k_tvl1_warp_lds grown to the step kernel's tile),
//   2. warps the tile's 8 rows per thread from it, two pixels at a time as packed float2 math — I1wx, I1wy, rho_c land in the registers the iterations read them from,
//      the same pixel taking the same bits whichever tile (owner or halo) computes it: warp_finish on the same taps,
//   3. re-uses the LDS for the iteration's neighbour planes and runs the head of the loop (TVL1_HEAD_ITERS = 2 iterations,
//      2-pixel halo: a tile owns 60 x 28 of its 64 x 32 pixels) with the packed tile function of the step kernel
//      (tvl1_tile.h) — same functions, same operation order, bit-identical,
//   4. stores u / p of the owned region into the other ping-pong set AND I1wx / I1wy / rho_c (a loop that goes on reads
//      them in the step kernel), publishes its share of the convergence sum; the last workgroup of the pair advances the
//      state machine (tvl1_ctrl.h: tvl1_plan_head / tvl1_end_head).
// Per owned pixel: 14.8 words read (7 planes x 1.22 + the image tiles' 6.3; 10 for a level's first warp, whose dual planes
// are zero by definition and are not read) + 9 written, against 31.3 for the two launches.  The kernel issues VALU
// instructions 3/4 of the time (SQ counters, profiles/round6/): forming I1x / I1y in LDS from a wider I1 tile instead of
// reading them (16 KB instead of 42 KB per tile) was measured SLOWER for that reason, as were hand-scheduled tap loads
// (LABNOTES.md section 11).
// Compiled with -ffp-contract=off (see tvl1_math.h).
#include <hip/hip_runtime.h>

#include "dfx_device.h"
#include "tvl1_device_common.h"
#include "tvl1_kernels.h"
#include "tvl1_tile.h"

#ifndef DFX_HEAD_P_EARLY
#define DFX_HEAD_P_EARLY 0 // the dual planes' loads issued in front of the warp (A/B builds: scripts/build_variant.sh)
#endif

namespace {

constexpr int HD_TW = 64, HD_TH = 32, HD_NW = 4, HD_K = TVL1_HEAD_ITERS;
// image tile: horizontal halo 8 (tile columns start at multiples of 60 -> 16-byte aligned row loads), vertical MARGIN + 2
constexpr int HD_MARGIN = 4, HD_HX = 8, HD_HY = HD_MARGIN + 2;
constexpr int HD_TWL = HD_TW + 2 * HD_HX, HD_THL = HD_TH + 2 * HD_HY;
constexpr int HD_LDS_IMG = 3 * HD_THL * HD_TWL;                       // floats: 10 560 (42 240 B)
constexpr int HD_LDS_ITER = (Q_PLANES * HD_TH + 4 * HD_NW) * HD_TW;   // floats:  9 216 (36 864 B)
static_assert(HD_LDS_ITER <= HD_LDS_IMG, "the iteration planes re-use the image tile's LDS");
static_assert((HD_TW - 2 * HD_K) % 4 == 0, "tile columns must start 16-byte aligned");
static_assert(HD_MARGIN + 2 <= HD_HX, "the bicubic window must fit the horizontal halo");

// Steps 1 + 2 for one thread: the warp of its HP float2 rows into pf[0..2] (I1wx, I1wy, rho_c), u1 / u2 into pf[3..4].
template <bool INTERIOR>
__device__ __forceinline__ void head_warp(const Tvl1LevelCtx &c, int b, int cur, int x0, int y0,
                                          float *tile, float (&pf)[PF_PLANES][HD_TH / HD_NW / 2][2]) {
    constexpr int HP = HD_TH / HD_NW / 2;
    using RM = RowMap<HD_TH, HD_NW>;
    const int lane = threadIdx.x & 63, role = RM::who();
    const int x = x0 + lane;
    const bool col_in = INTERIOR || (x >= 0 && x < c.w);
    const PairDesc pd = c.pairs[b];
    const float *I0 = c.frame_I + (long long)pd.frame_a * c.frame_stride + c.lvl_off;
    const long long fb = (long long)pd.frame_b * c.frame_stride + c.lvl_off;
    const float *P1 = c.frame_I + fb, *P1x = c.frame_Ix + fb, *P1y = c.frame_Iy + fb;
    // (buffer addressing, tvl1_device_common.h: the pair's slot, and a descriptor on this level's plane of frame a)
    const dfx_rsrc rs = pair_rsrc(c, b), r0 = dfx_make_rsrc(I0, 4u * (unsigned)(c.pitch * c.h));
    const unsigned s_u1 = plane_soff(c, PL_U1_0 + 2 * cur), s_u2 = plane_soff(c, PL_U2_0 + 2 * cur);
    float i0r[HP][2];
#pragma unroll
    for (int j = 0; j < HP; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int y = y0 + RM::row(role, j, e);
            const bool in = INTERIOR || (col_in && y >= 0 && y < c.h);
            const unsigned o = in ? 4u * (unsigned)(y * c.pitch + x) : 0u; // masked lanes read element 0
            pf[3][j][e] = buf_ld(rs, o, s_u1);
            pf[4][j][e] = buf_ld(rs, o, s_u2);
            i0r[j][e] = buf_ld(r0, o, 0u);
        }
    const int tx0 = x0 - HD_HX, ty0 = y0 - HD_HY;
    // LDS image tile: (I1, I1x) of a pixel side by side — a tap of the bicubic sums is one ds_read_b64 for both (2 LDS cycles per
    // wave instead of 4: the warp phase keeps the LDS nearly as busy as the vector ALU) — and I1y as a plane of its own
    f2 *t1x = reinterpret_cast<f2 *>(tile);
    float *tgy = tile + 2 * HD_THL * HD_TWL;
    // image tiles: float4 q of tile row r covers image columns tx0 + 4q .. + 3 of row clamp(ty0 + r); clamp-to-edge is
    // applied here, so a tile entry IS the point-sampled texture value (k_tvl1_warp_lds has the same copy loop): a float4
    // left of the image is column 0 four times, one right of it column w - 1, one that straddles the right border is patched
    // per element (the row pitch is a multiple of 64 floats >= w, so the aligned 16-byte load never leaves the row).
    constexpr int Q = HD_TWL / 4;
    for (int i = threadIdx.x; i < HD_THL * Q; i += 64 * HD_NW) {
        const int r = i / Q, q = i - r * Q;
        const long long ro = (long long)min(max(ty0 + r, 0), c.h - 1) * c.pitch;
        const int gx = tx0 + 4 * q;
        const int lx4 = min(max(gx, 0), ((c.w - 1) >> 2) << 2); // aligned, inside the row
        // (plain pointers: this compiler's __builtin_amdgcn_raw_buffer_load_b128 emits a ONE-dword load and splats it)
        const float4 a = *reinterpret_cast<const float4 *>(P1 + ro + lx4);
        const float4 bq = *reinterpret_cast<const float4 *>(P1x + ro + lx4);
        const float4 cq = *reinterpret_cast<const float4 *>(P1y + ro + lx4);
        const int o = r * HD_TWL + 4 * q;
        if (gx >= 0 && gx + 3 <= c.w - 1) { // (the image tile is wider than the iteration tile INTERIOR speaks about)
            *reinterpret_cast<float4 *>(&t1x[o]) = make_float4(a.x, bq.x, a.y, bq.y);
            *reinterpret_cast<float4 *>(&t1x[o + 2]) = make_float4(a.z, bq.z, a.w, bq.w);
            *reinterpret_cast<float4 *>(&tgy[o]) = cq;
            continue;
        }
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w}, cv[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = min(max(gx + e, 0), c.w - 1) - lx4; // 0..3: the element of the loaded float4 column gx + e clamps to
            t1x[o + e] = pk_set(k == 0 ? av[0] : k == 1 ? av[1] : k == 2 ? av[2] : av[3],
                                k == 0 ? bv[0] : k == 1 ? bv[1] : k == 2 ? bv[2] : bv[3]);
            tgy[o + e] = k == 0 ? cv[0] : k == 1 ? cv[1] : k == 2 ? cv[2] : cv[3];
        }
    }
    __syncthreads();
    // Two pixels at a time — the rows of float2 j — with the 16 x 3 multiply-adds of the bicubic sums as packed
    // operations: each half performs warp_finish's very sequence of rounded operations (packed f32 operations round
    // each half on its own, nothing is contracted), the final 1 / wsum and the rho_c chain run per half in scalar form.
    // A pixel whose 4 x 4 window leaves the LDS tile (|flow| beyond the margin: occlusion borders, large motion) is
    // noted in `far` and redone with the global gather afterwards (rare; kept out of this loop's register budget).
    unsigned far = 0u;
    const float xf = (float)x;
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        const int ya = y0 + RM::row(role, j, 0), yb = y0 + RM::row(role, j, 1);
        const bool ina = INTERIOR || (col_in && ya >= 0 && ya < c.h), inb = INTERIOR || (col_in && yb >= 0 && yb < c.h);
        const f2 u1v = pk_set(pf[3][j][0], pf[3][j][1]), u2v = pk_set(pf[4][j][0], pf[4][j][1]);
        const f2 wx = (f2)(xf) + u1v, wy = pk_set((float)ya, (float)yb) + u2v;
        const f2 fx0 = __builtin_elementwise_ceil(wx - 2.0f), fy0 = __builtin_elementwise_ceil(wy - 2.0f);
        // the window's first tap, exactly as warp_fetch derives it (clamped before the int conversion: NaN / Inf flows)
        const int lxa = (int)fminf(fmaxf(fx0.x, -4.0f), (float)c.w + 4.0f) - tx0;
        const int lxb = (int)fminf(fmaxf(fx0.y, -4.0f), (float)c.w + 4.0f) - tx0;
        const int lya = (int)fminf(fmaxf(fy0.x, -4.0f), (float)c.h + 4.0f) - ty0;
        const int lyb = (int)fminf(fmaxf(fy0.y, -4.0f), (float)c.h + 4.0f) - ty0;
        // (0 <= l && l + 3 < N as one unsigned comparison)
        const bool oka = (unsigned)lxa < (unsigned)(HD_TWL - 3) && (unsigned)lya < (unsigned)(HD_THL - 3);
        const bool okb = (unsigned)lxb < (unsigned)(HD_TWL - 3) && (unsigned)lyb < (unsigned)(HD_THL - 3);
        far |= (ina && !oka ? 1u : 0u) << (2 * j);
        far |= (inb && !okb ? 1u : 0u) << (2 * j + 1);
        const int oa = oka ? lya * HD_TWL + lxa : 0, ob = okb ? lyb * HD_TWL + lxb : 0; // (outside the tile: redone below)
        // every weight's arm chosen by the tap's position in the window (pk_bicubic_window: half the instructions of the
        // select chain); a pixel for which that is not the chain's choice (one-ulp coincidences, NaN / infinite flows) is
        // redone below with the scalar chain, like one whose window leaves the tile
        f2 cwx[4], cwy[4];
        bool wxa, wxb, wya, wyb;
        pk_bicubic_window(wx, fx0, cwx, wxa, wxb);
        pk_bicubic_window(wy, fy0, cwy, wya, wyb);
        far |= (ina && !(wxa && wya) ? 1u : 0u) << (2 * j);
        far |= (inb && !(wxb && wyb) ? 1u : 0u) << (2 * j + 1);
        // (I1, I1x) sums per pixel (sa: pixel a, sb: pixel b; the weight enters as one half of wgt, by op_sel), I1y sums and
        // the weight sum for both pixels at once: per value the same products and additions in the same order as before
        f2 sa = (f2)(0.0f), sb = (f2)(0.0f), sumy = (f2)(0.0f), wsum = (f2)(0.0f);
#pragma unroll
        for (int jy = 0; jy < 4; ++jy) {
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const int o = jy * HD_TWL + jx;
                const f2 wgt = cwx[jx] * cwy[jy];
                sa = sa + pk_set(wgt.x, wgt.x) * t1x[oa + o];
                sb = sb + pk_set(wgt.y, wgt.y) * t1x[ob + o];
                sumy = sumy + wgt * pk_set(tgy[oa + o], tgy[ob + o]);
                wsum = wsum + wgt;
            }
        }
        const f2 sum = pk_set(sa.x, sb.x), sumx = pk_set(sa.y, sb.y);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float coeff = 1.0f / (e ? wsum.y : wsum.x);
            const float I1w = (e ? sum.y : sum.x) * coeff;
            const float I1wx = (e ? sumx.y : sumx.x) * coeff, I1wy = (e ? sumy.y : sumy.x) * coeff;
            const float rho_c = ((I1w - I1wx * pf[3][j][e]) - I1wy * pf[4][j][e]) - i0r[j][e];
            const bool in = e ? inb : ina;
            pf[0][j][e] = in ? I1wx : 0.0f;
            pf[1][j][e] = in ? I1wy : 0.0f;
            pf[2][j][e] = in ? rho_c : 0.0f;
        }
    }
    if (far) {
#pragma unroll
        for (int j = 0; j < HP; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (far & (1u << (2 * j + e))) {
                    const int y = y0 + RM::row(role, j, e);
                    const WarpOut r = warp_backward_px_v(P1, P1x, P1y, c.w, c.h, c.pitch, x, y, pf[3][j][e], pf[4][j][e], i0r[j][e]);
                    pf[0][j][e] = r.I1wx;
                    pf[1][j][e] = r.I1wy;
                    pf[2][j][e] = r.rho_c;
                }
    }
}

// the four dual planes of ping-pong set `cur` into pf[5..8]; p_zero (wave-uniform): the level's first warp — p = 0 (A.3),
// nobody has written those planes (k_tvl1_zero_planes leaves them alone when this kernel is in use)
template <bool INTERIOR>
__device__ __forceinline__ void head_load_p(const Tvl1LevelCtx &c, int b, int cur, int x0, int y0, bool p_zero,
                                            float (&pf)[PF_PLANES][HD_TH / HD_NW / 2][2]) {
    constexpr int HP = HD_TH / HD_NW / 2;
    if (p_zero) {
#pragma unroll
        for (int j = 0; j < HP; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pf[5 + q][j][e] = 0.0f;
        return;
    }
    using RM = RowMap<HD_TH, HD_NW>;
    const int lane = threadIdx.x & 63, role = RM::who();
    const int x = x0 + lane;
    const bool col_in = INTERIOR || (x >= 0 && x < c.w);
    const dfx_rsrc rs = pair_rsrc(c, b);
    const unsigned so[4] = {plane_soff(c, PL_P11_0 + 4 * cur), plane_soff(c, PL_P12_0 + 4 * cur),
                            plane_soff(c, PL_P21_0 + 4 * cur), plane_soff(c, PL_P22_0 + 4 * cur)};
#pragma unroll
    for (int j = 0; j < HP; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int y = y0 + RM::row(role, j, e);
            const bool in = INTERIOR || (col_in && y >= 0 && y < c.h);
            const unsigned o = in ? 4u * (unsigned)(y * c.pitch + x) : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pf[5 + q][j][e] = buf_ld(rs, o, so[q]);
        }
}

// I1wx / I1wy / rho_c of the owned region (tile_store's ownership rule): a loop that goes on reads them in the step kernel
template <bool INTERIOR>
__device__ __forceinline__ void head_store_warp(const Tvl1LevelCtx &c, int b, int x0, int y0,
                                                const TileState<HD_TH / HD_NW / 2> &T, bool own_lo, bool own_hi) {
    constexpr int HP = HD_TH / HD_NW / 2, K = HD_K;
    using RM = RowMap<HD_TH, HD_NW>;
    const int lx = threadIdx.x & 63, role = RM::who();
    const int gx = x0 + lx;
    const bool col_in = INTERIOR || (gx >= 0 && gx < c.w);
    const bool col_owned = (lx >= K || own_lo) && (lx < HD_TW - K || own_hi) && col_in;
    const dfx_rsrc rs = pair_rsrc(c, b);
    const unsigned s_wx = plane_soff(c, PL_I1WX), s_wy = plane_soff(c, PL_I1WY), s_rc = plane_soff(c, PL_RHOC);
#pragma unroll
    for (int j = 0; j < HP; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ly = RM::row(role, j, e), gy = y0 + ly;
            if (col_owned && ly >= K && ly < HD_TH - K && (INTERIOR || (gy >= 0 && gy < c.h))) {
                const unsigned o = 4u * (unsigned)(gy * c.pitch + gx);
                buf_st(rs, o, s_wx, e ? T.kwx[j].y : T.kwx[j].x);
                buf_st(rs, o, s_wy, e ? T.kwy[j].y : T.kwy[j].x);
                buf_st(rs, o, s_rc, e ? T.krc[j].y : T.krc[j].x);
            }
        }
}

template <bool INTERIOR, int MATH>
__device__ __forceinline__ double head_tile(const Tvl1LevelCtx &c, int b, float *lds_raw, const Tvl1StepPlan &plan, int x0,
                                            int y0, bool own_lo, bool own_hi, bool p_zero) {
    constexpr int TH = HD_TH, NW = HD_NW, HP = TH / NW / 2;
    float (*lds)[TH][HD_TW] = reinterpret_cast<float (*)[TH][HD_TW]>(lds_raw);
    float (*bnd)[2 * NW][HD_TW] = reinterpret_cast<float (*)[2 * NW][HD_TW]>(lds_raw + Q_PLANES * TH * HD_TW);
    float pf[PF_PLANES][HP][2];
    TileState<HP> T;
    const int role = RowMap<TH, NW>::who();
#if DFX_HEAD_P_EARLY
    head_load_p<INTERIOR>(c, b, plan.src, x0, y0, p_zero, pf); // in flight while the warp runs
#endif
    head_warp<INTERIOR>(c, b, plan.src, x0, y0, lds_raw, pf);
#if !DFX_HEAD_P_EARLY
    head_load_p<INTERIOR>(c, b, plan.src, x0, y0, p_zero, pf);
#endif
    __syncthreads(); // every thread is done with the image tile: its LDS becomes the iteration's neighbour planes
    tile_consume<TH, NW, INTERIOR, MATH>(c, x0, y0, pf, T, lds, bnd);
    __syncthreads();
    double dsum;
    if (role * HP < HD_K) // only roles that hold halo rows carry the per-float2 skip tests
        dsum = tile_iterate_trap<TH, NW, INTERIOR, true, MATH>(c, T, lds, bnd, plan.n_iters, plan.do_check != 0, HD_K, x0, y0,
                                                               role, own_lo, own_hi);
    else
        dsum = tile_iterate_trap<TH, NW, INTERIOR, false, MATH>(c, T, lds, bnd, plan.n_iters, plan.do_check != 0, HD_K, x0,
                                                                y0, role, own_lo, own_hi);
    tile_store<TH, NW, INTERIOR>(c, b, plan.src ^ 1, HD_K, x0, y0, T, own_lo, own_hi);
    head_store_warp<INTERIOR>(c, b, x0, y0, T, own_lo, own_hi);
    return dsum;
}

} // namespace

// One workgroup = one 64 x 32 tile of one pair; grid.x = tiles of the head's geometry (tvl1_head_blocks), grid.z = pair.
template <int MATH>
__global__ __launch_bounds__(64 * HD_NW, 3) void k_tvl1_warp_head(Tvl1LevelCtx c, int step_id) {
    __shared__ __attribute__((aligned(16))) float lds_raw[HD_LDS_IMG];
    __shared__ double lds_red[8];
    __shared__ int lds_flag;
    const int b = blockIdx.z;
    Tvl1State *st = c.state + b;
    if (st->phase != TVL1_PH_WARP)
        return;
    // every workgroup derives the head's plan from the state as it stands (it changes only once every workgroup of the
    // pair has arrived, below)
    const Tvl1StepPlan plan = tvl1_plan_head(*st, c.loop);
    const bool p_zero = st->warp == 0; // the level's first warp: p = 0 by definition, not by reading zeros
    const unsigned nblk = gridDim.x;
    const Tvl1StepGeom g = tvl1_step_geom(c.w, c.h, HD_TW, HD_TH, HD_K, 1);
    const Tvl1TilePlace tp = tvl1_tile_place(g, HD_TW, HD_TH, dfx_xcd_tile_index((int)blockIdx.x, (int)nblk));
    const int xs = tp.x0, ys = tp.y0;
    const bool interior = xs >= 1 && ys >= 1 && xs + HD_TW + 1 <= c.w && ys + HD_TH + 1 <= c.h;
    double dsum;
    if (interior)
        dsum = head_tile<true, MATH>(c, b, lds_raw, plan, xs, ys, false, false, p_zero);
    else
        dsum = head_tile<false, MATH>(c, b, lds_raw, plan, xs, ys, tp.own_lo != 0, tp.own_hi != 0, p_zero);

    // the tile's share of sum(diff), the arrival ticket, and — in the pair's last workgroup — the state transition
    const int tid = threadIdx.x;
    double *partials = c.partials + (long long)b * c.partials_stride;
    if (plan.do_check) {
        const double bs = block_reduce_sum_f64(dsum, lds_red);
        if (tid == 0)
            publish_partial(partials + blockIdx.x, bs);
    }
    if (!arrive_is_last(st, nblk, &lds_flag))
        return;
    double err = 0.0;
    if (plan.do_check) {
        double acc = 0.0;
        for (unsigned i = tid; i < nblk; i += blockDim.x)
            acc += read_partial(partials + i);
        err = block_reduce_sum_f64(acc, lds_red);
    }
#if DFX_TVL1_DEBUG // measurement builds: never converge, so every build runs the same step schedule
    err = 1e300;
#endif
    if (tid == 0) {
        tvl1_end_head(*st, c.loop, plan, step_id, err);
        if (st->phase == TVL1_PH_LEVEL_DONE)
            finish_level(c, b, *st, step_id);
        __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, AGENT);
    }
}

int tvl1_head_blocks(const Tvl1LevelCtx &c) { return tvl1_step_grid(c.w, c.h, HD_TW, HD_TH, HD_K, 1, 1); }

void tvl1_launch_warp_head(hipStream_t s, const Tvl1LevelCtx &c, int step_id, int math) {
    const dim3 grid(tvl1_head_blocks(c), 1, c.n_pairs), block(64 * HD_NW);
    if (math == 1)
        hipLaunchKernelGGL((k_tvl1_warp_head<1>), grid, block, 0, s, c, step_id);
    else if (math == TVL1_HYP_SQRT)
        hipLaunchKernelGGL((k_tvl1_warp_head<TVL1_HYP_SQRT>), grid, block, 0, s, c, step_id);
    else if (math == TVL1_HYP_LIBM)
        hipLaunchKernelGGL((k_tvl1_warp_head<TVL1_HYP_LIBM>), grid, block, 0, s, c, step_id);
    else
        hipLaunchKernelGGL((k_tvl1_warp_head<0>), grid, block, 0, s, c, step_id);
}
