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
DFX_HD void tvl1_end_segment(Tvl1State &s, const Tvl1LoopCfg &c, const Tvl1StepPlan &p, int step_id, double error) {
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
