// tvl1_ctrl.h — per-pair control state of the TVL1 inner loop, advanced ON THE DEVICE.
//
// cv::cuda::OpticalFlowDual_TVL1 drives its inner loop from the host: every convergence check is a
// stream sync + a reduction read back (reference call site src/denseflow_gpu.cpp:327; loop shape
// SURVEY.md A.3/A.4).  Here the same loop is a small state machine that lives in HBM next to the
// planes it controls.  The host only enqueues identical "step" launches; each step looks at the
// state, does the work that is due (a warp, or up to K fused inner iterations) and the last
// workgroup to finish advances the state.  No host round trip inside a pair.
//
// The functions below are plain C++ (no HIP) so the exact same code is compiled into the kernels
// and into the CPU unit tests that replay the oracle's error traces through it.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define DFX_HD __host__ __device__ __forceinline__
#else
#define DFX_HD inline
#endif

enum : int { TVL1_PH_WARP = 0, TVL1_PH_ITER = 1, TVL1_PH_LEVEL_DONE = 2 };

#define TVL1_MAX_WARPS 16
#define TVL1_NO_CHECK 0x3fffffff

struct Tvl1State {
    int phase;        // TVL1_PH_*
    int warp;         // warp index in [0, warps)
    int cur;          // ping-pong set holding the current u/p at seg_step0
    int seg_step0;    // id of the first step of the current segment
    int seg_n0;       // inner-iteration index executed first in the current segment
    int next_check;   // iteration index whose estimateU evaluates the error (TVL1_NO_CHECK: none left)
    int n_checks;     // statistics: convergence sums evaluated at this level
    int pad0_;
    double prev_error;
    double thr;       // scaledEpsilon = eps^2 * W*H of the level
    int iters[TVL1_MAX_WARPS]; // statistics: inner iterations executed per warp at this level
    unsigned int ticket;       // arrival counter of the current step's workgroups
    int pad_;
    // statistics: tile-row updates executed per tile column at this level, in half rows (a primal or a dual update of one
    // tile row = 1), by the step kernel's tiles and by the warp-and-head kernel's (tvl1_step_work)
    long long step_work, head_work;
};

struct Tvl1LoopCfg {
    int warps;
    int iterations;
    int fuse_k; // inner iterations per step (>=1)
};

// A.4: first iteration index m >= n_start at which calcError is true, replaying
//   calcError = (n & 1) && (prevError < thr);   else prevError -= thr
// prev_error is updated to the value it has when that check is reached.
DFX_HD int tvl1_next_check(int n_start, int iterations, double thr, double *prev_error) {
    double pe = *prev_error;
    for (int m = n_start; m < iterations; ++m) {
        if ((m & 1) && (pe < thr)) {
            *prev_error = pe;
            return m;
        }
        pe -= thr;
    }
    *prev_error = pe;
    return TVL1_NO_CHECK;
}

// Last iteration index of the current segment: the next check, or the loop bound.
DFX_HD int tvl1_segment_end(const Tvl1State &s, const Tvl1LoopCfg &c) {
    return s.next_check < c.iterations ? s.next_check : c.iterations - 1;
}

// What a step has to do in phase ITER.  Returns the number of iterations (0 = nothing).
struct Tvl1StepPlan {
    int n_first, n_iters; // iterations [n_first, n_first + n_iters)
    int src;              // ping-pong set read by this step (dst = src ^ 1)
    int is_last;          // this step ends the segment -> ticket protocol + state advance
    int do_check;         // the last iteration of this step evaluates the error sum
};

DFX_HD Tvl1StepPlan tvl1_plan_step(const Tvl1State &s, const Tvl1LoopCfg &c, int step_id) {
    Tvl1StepPlan p;
    const int j = step_id - s.seg_step0;
    const int end_n = tvl1_segment_end(s, c);
    p.n_first = s.seg_n0 + j * c.fuse_k;
    int last = p.n_first + c.fuse_k - 1;
    if (last > end_n)
        last = end_n;
    p.n_iters = last - p.n_first + 1;
    if (j < 0 || p.n_iters < 0)
        p.n_iters = 0;
    p.src = s.cur ^ (j & 1);
    p.is_last = (last == end_n) && p.n_iters > 0;
    p.do_check = p.is_last && (end_n == s.next_check);
    return p;
}

// Work of one step of n fused iterations on a 32-row tile with a K-row halo, in half rows (primal / dual update of one
// tile row): 64 per iteration minus what the trapezoid layout skips (tvl1_tile.h: with d iterations to go the primal
// update is skipped on rows closer than K - d - 1 to the tile's top / bottom edge, the dual update on rows closer than
// K - d).  bench.py's useful_frac = owned pixel-iterations / (this x 32 lanes x tiles).
DFX_HD int tvl1_step_work(int n, int K) {
    int w = 0;
    for (int it = 0; it < n; ++it) {
        const int need = K - (n - 1 - it);
        const int sp = need - 1 < 0 ? 0 : (need - 1 > 16 ? 16 : need - 1), sd = need < 0 ? 0 : (need > 16 ? 16 : need);
        w += 64 - 2 * sp - 2 * sd;
    }
    return w;
}

// Start the inner loop of a warp (called when the warp's backward warping has been done at step_id).
DFX_HD void tvl1_begin_loop(Tvl1State &s, const Tvl1LoopCfg &c, int step_id) {
    s.phase = TVL1_PH_ITER;
    s.seg_step0 = step_id + 1;
    s.seg_n0 = 0;
    s.prev_error = 0.0; // error = DBL_MAX, prevError = 0 (A.3)
    s.next_check = tvl1_next_check(0, c.iterations, s.thr, &s.prev_error);
    if (c.iterations <= 0) { // degenerate: no inner iterations at all
        s.iters[s.warp] = 0;
        s.warp += 1;
        s.phase = (s.warp < c.warps) ? TVL1_PH_WARP : TVL1_PH_LEVEL_DONE;
    }
}

// Advance after the segment-final step `step_id` (plan p, error = sum(diff) if p.do_check).
DFX_HD void tvl1_end_segment_core(Tvl1State &s, const Tvl1LoopCfg &c, const Tvl1StepPlan &p, int step_id, double error);
DFX_HD void tvl1_end_segment(Tvl1State &s, const Tvl1LoopCfg &c, const Tvl1StepPlan &p, int step_id, double error) {
    // the segment's steps: full ones of fuse_k iterations and the last one (statistics)
    const int len = p.n_first + p.n_iters - s.seg_n0, last = p.n_iters;
    s.step_work += (long long)((len - last) / c.fuse_k) * tvl1_step_work(c.fuse_k, c.fuse_k) + tvl1_step_work(last, c.fuse_k);
    tvl1_end_segment_core(s, c, p, step_id, error);
}
DFX_HD void tvl1_end_segment_core(Tvl1State &s, const Tvl1LoopCfg &c, const Tvl1StepPlan &p, int step_id, double error) {
    const int end_n = p.n_first + p.n_iters - 1;
    bool loop_done;
    if (p.do_check) {
        s.n_checks += 1;
        // for (...; error > scaledEpsilon && n < iterations; ++n)
        loop_done = !(error > s.thr) || (end_n + 1 >= c.iterations);
        if (!loop_done) {
            s.prev_error = error;
            s.next_check = tvl1_next_check(end_n + 1, c.iterations, s.thr, &s.prev_error);
        }
    } else {
        loop_done = true; // ran into the iteration bound without a further check
    }
    s.cur = p.src ^ 1; // the set this step wrote
    if (loop_done) {
        s.iters[s.warp] = end_n + 1;
        s.warp += 1;
        s.phase = (s.warp < c.warps) ? TVL1_PH_WARP : TVL1_PH_LEVEL_DONE;
    } else {
        s.seg_step0 = step_id + 1;
        s.seg_n0 = end_n + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// The warp-and-head kernel (k_tvl1_warp_head, tvl1_head_kernels.hip): the backward warp of a pair AND the head of the inner
// loop it starts — its first TVL1_HEAD_ITERS iterations, which on converging content are the whole loop: the first
// convergence check is due at iteration 1 (tvl1_next_check(0, ...)), and warps 1-4 of a level usually pass it — in ONE
// launch, in front of the step kernel of the same step id.  I1wx / I1wy / rho_c go from the warp into the iterations in
// registers.  These two functions are the state transitions of that launch; the CPU harness replays them
// (tests/ctrl_harness.cpp, split_warp = 2).
#define TVL1_HEAD_ITERS 2

// The plan of the head, derived by every workgroup from the state as the pair's warp phase finds it (read-only: the state
// changes once, in tvl1_end_head): what tvl1_begin_loop + tvl1_plan_step would plan for a first step of TVL1_HEAD_ITERS
// iterations — iterations [0, min(TVL1_HEAD_ITERS, segment length)) reading set s.cur.  Requires c.iterations >= 1.
DFX_HD Tvl1StepPlan tvl1_plan_head(const Tvl1State &s, const Tvl1LoopCfg &c) {
    double pe = 0.0;
    const int next_check = tvl1_next_check(0, c.iterations, s.thr, &pe);
    const int end_n = next_check < c.iterations ? next_check : c.iterations - 1;
    Tvl1StepPlan p;
    p.n_first = 0;
    const int last = TVL1_HEAD_ITERS - 1 < end_n ? TVL1_HEAD_ITERS - 1 : end_n;
    p.n_iters = last + 1;
    p.src = s.cur;
    p.is_last = last == end_n;
    p.do_check = p.is_last && end_n == next_check;
    return p;
}

// The state transition of the launch (one thread, once every workgroup of the pair has arrived): the loop begins and its
// head is accounted for.  Segment finished: exactly tvl1_end_segment, a further segment starting at THIS step id (the step
// kernel follows in the stream).  Segment not finished (no check falls into the head: epsilon = 0): the same segment
// goes on at iteration n_iters from the set the head wrote.
DFX_HD void tvl1_end_head(Tvl1State &s, const Tvl1LoopCfg &c, const Tvl1StepPlan &p, int step_id, double error) {
    tvl1_begin_loop(s, c, step_id - 1);
    s.head_work += tvl1_step_work(p.n_iters, TVL1_HEAD_ITERS);
    if (p.is_last) {
        tvl1_end_segment_core(s, c, p, step_id - 1, error);
    } else {
        s.cur = p.src ^ 1;
        s.seg_n0 = p.n_first + p.n_iters;
        s.seg_step0 = step_id;
    }
}

// ------------------------------------------------------------------------------------------------
// Tile geometry of a step of the default step kernel (k_tvl1_step_fused, 64 x 32 tiles with a K-pixel halo).
//   shift = 1 (default): tile columns start at x = 0 instead of -K: the first tile also owns its left halo columns (the
//               image border needs no halo) and the last one everything up to the right border — ceil((w - 2K) /
//               (64 - 2K)) tile columns instead of ceil(w / (64 - 2K)): 14 instead of 15 at the coarsest 1080p level;
//   shift = 0: the classic tiling (kept as a cross-check: dfx_params.variant & DFX_VAR_TVL1_CLASSIC_GEOM).
// The kernel, the launcher (grid size) and the CPU test of the ownership partition all use these functions.
struct Tvl1StepGeom {
    int halo;     // K
    int ntx, nty; // tiles of a step
    int shift;
};

DFX_HD Tvl1StepGeom tvl1_step_geom(int w, int h, int tw, int th, int K, int shift) {
    Tvl1StepGeom g;
    g.halo = K;
    g.shift = shift ? 1 : 0;
    const int sw = tw - 2 * g.halo, sh = th - 2 * g.halo;
    if (g.shift) {
        const int n = (w - 2 * g.halo + sw - 1) / sw;
        g.ntx = n > 1 ? n : 1;
    } else {
        g.ntx = (w + sw - 1) / sw;
    }
    g.nty = (h + sh - 1) / sh;
    return g;
}

struct Tvl1TilePlace {
    int x0, y0;         // image coordinates of the tile's lane 0 / row 0 (may be negative)
    int own_lo, own_hi; // the tile also owns its left / right halo columns (image border)
};

DFX_HD Tvl1TilePlace tvl1_tile_place(const Tvl1StepGeom &g, int tw, int th, int tile) {
    Tvl1TilePlace p;
    const int ty = tile / g.ntx, tx = tile - ty * g.ntx;
    const int sw = tw - 2 * g.halo, sh = th - 2 * g.halo;
    p.x0 = g.shift ? tx * sw : tx * sw - g.halo;
    p.y0 = ty * sh - g.halo;
    p.own_lo = g.shift && tx == 0;
    p.own_hi = g.shift && tx == g.ntx - 1;
    return p;
}

// Is lane lx / row ly of a placed tile written back by this tile?  (Rows and columns outside the image are not.)
DFX_HD bool tvl1_tile_owns(const Tvl1StepGeom &g, const Tvl1TilePlace &p, int tw, int th, int lx, int ly, int w, int h) {
    const int gx = p.x0 + lx, gy = p.y0 + ly;
    return (lx >= g.halo || p.own_lo) && (lx < tw - g.halo || p.own_hi) && ly >= g.halo && ly < th - g.halo &&
           gx >= 0 && gx < w && gy >= 0 && gy < h;
}

// Workgroups per pair of a step launch = tiles of the step.  With an in-kernel warp phase (which tiles the image the
// classic way) the classic count is needed.
DFX_HD int tvl1_step_grid(int w, int h, int tw, int th, int K, int shift, int split_warp) {
    const Tvl1StepGeom g = tvl1_step_geom(w, h, tw, th, K, split_warp ? shift : 0);
    return g.ntx * g.nty;
}
