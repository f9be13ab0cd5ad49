// tests/ctrl_harness.cpp — TEST INFRASTRUCTURE.  Drives the device-side TVL1 control state machine
// (denseflow_amd/csrc/tvl1_ctrl.h, the very header compiled into the kernels) on the CPU, using the
// oracle's stage functions as the "kernels", so the segment/check/step bookkeeping can be compared
// with the oracle's plain host loop for any fuse_k without a GPU.
#include <cstring>
#include <vector>

#include "../denseflow_amd/csrc/tvl1_ctrl.h"
#include "../oracle/tvl1_oracle.h"

// split_warp = 0: a warp occupies a step of its own (the warp phase inside the step kernel);
// split_warp = 1: the dedicated warp kernel runs in front of the step kernel of the SAME step id (k_tvl1_warp): the warp
//                 starts the loop at this very step, and with zero iterations the step kernel skips the pair;
// split_warp = 2: that kernel also runs the head of the loop (k_tvl1_warp_head: tvl1_plan_head / tvl1_end_head); with
//                 zero iterations the engine falls back to split_warp = 1, and so does this replay.
extern "C" int ctrl_replay_level_ex(const float *I0, const float *I1, float *u1, float *u2, int W, int H, int warps,
                                    int iterations, int fuse_k, double eps, double lambda, double theta, double tau,
                                    int *iters_out, int *n_checks_out, int *steps_out, int split_warp);

extern "C" int ctrl_replay_level(const float *I0, const float *I1, float *u1, float *u2, int W, int H, int warps,
                                 int iterations, int fuse_k, double eps, double lambda, double theta, double tau,
                                 int *iters_out, int *n_checks_out, int *steps_out) {
    return ctrl_replay_level_ex(I0, I1, u1, u2, W, H, warps, iterations, fuse_k, eps, lambda, theta, tau, iters_out,
                                n_checks_out, steps_out, 0);
}

extern "C" int ctrl_replay_level_ex(const float *I0, const float *I1, float *u1, float *u2, int W, int H, int warps,
                                    int iterations, int fuse_k, double eps, double lambda, double theta, double tau,
                                    int *iters_out, int *n_checks_out, int *steps_out, int split_warp) {
    const size_t n = (size_t)W * H;
    std::vector<float> buf(n * 11, 0.f);
    float *I1x = buf.data(), *I1y = I1x + n, *I1w = I1y + n, *I1wx = I1w + n, *I1wy = I1wx + n, *grad = I1wy + n,
          *rho = grad + n, *p11 = rho + n, *p12 = p11 + n, *p21 = p12 + n, *p22 = p21 + n;
    orc_tvl1_centered_gradient(I1, W, H, I1x, I1y);
    const float l_t = (float)(lambda * theta), taut = (float)(tau / theta), th = (float)theta;

    Tvl1LoopCfg cfg{warps, iterations, fuse_k};
    Tvl1State st;
    std::memset(&st, 0, sizeof st);
    st.phase = warps > 0 ? TVL1_PH_WARP : TVL1_PH_LEVEL_DONE;
    st.thr = eps * eps * (double)(W * H);
    st.next_check = TVL1_NO_CHECK;

    int step_id = 0;
    const int limit = warps * (iterations + 2) + 64;
    for (; st.phase != TVL1_PH_LEVEL_DONE && step_id < limit; ++step_id) {
        if (st.phase == TVL1_PH_WARP && split_warp == 2 && iterations >= 1) {
            orc_tvl1_warp_backward(I0, I1, I1x, I1y, u1, u2, W, H, I1w, I1wx, I1wy, grad, rho);
            const Tvl1StepPlan hp = tvl1_plan_head(st, cfg);
            { // the read-only plan is what begin_loop + plan_step plan for a first step of TVL1_HEAD_ITERS iterations
                Tvl1State t = st;
                Tvl1LoopCfg hc = cfg;
                hc.fuse_k = TVL1_HEAD_ITERS;
                tvl1_begin_loop(t, hc, step_id - 1);
                const Tvl1StepPlan q = tvl1_plan_step(t, hc, step_id);
                if (q.n_first != hp.n_first || q.n_iters != hp.n_iters || q.src != hp.src || q.is_last != hp.is_last ||
                    q.do_check != hp.do_check)
                    return -3;
            }
            if (hp.n_iters <= 0 || hp.n_iters > TVL1_HEAD_ITERS)
                return -4;
            double herr = 0.0;
            for (int k = 0; k < hp.n_iters; ++k) {
                const int check = hp.do_check && (k == hp.n_iters - 1);
                herr = orc_tvl1_estimate_u(I1wx, I1wy, grad, rho, p11, p12, p21, p22, u1, u2, W, H, l_t, th, check);
                orc_tvl1_estimate_dual(u1, u2, p11, p12, p21, p22, W, H, taut);
            }
            tvl1_end_head(st, cfg, hp, step_id, herr);
            if (st.phase != TVL1_PH_ITER)
                continue; // the step kernel of this step id skips pairs that are not iterating
        } else if (st.phase == TVL1_PH_WARP) {
            orc_tvl1_warp_backward(I0, I1, I1x, I1y, u1, u2, W, H, I1w, I1wx, I1wy, grad, rho);
            tvl1_begin_loop(st, cfg, split_warp ? step_id - 1 : step_id);
            if (!split_warp || st.phase != TVL1_PH_ITER)
                continue; // split: the step kernel of this step id follows; it skips pairs that are not iterating
        }
        const Tvl1StepPlan p = tvl1_plan_step(st, cfg, step_id);
        if (p.n_iters <= 0)
            return -2;
        double err = 0.0;
        for (int k = 0; k < p.n_iters; ++k) {
            const int check = p.do_check && (k == p.n_iters - 1);
            err = orc_tvl1_estimate_u(I1wx, I1wy, grad, rho, p11, p12, p21, p22, u1, u2, W, H, l_t, th, check);
            orc_tvl1_estimate_dual(u1, u2, p11, p12, p21, p22, W, H, taut);
        }
        if (p.is_last)
            tvl1_end_segment(st, cfg, p, step_id, err);
    }
    if (st.phase != TVL1_PH_LEVEL_DONE)
        return -1;
    for (int w = 0; w < warps; ++w)
        iters_out[w] = st.iters[w];
    *n_checks_out = st.n_checks;
    *steps_out = step_id;
    return 0;
}

// Ownership partition of a step's tile geometry (tvl1_step_geom / tvl1_tile_place / tvl1_tile_owns — the functions the
// step kernel and its launcher use).  Returns 0 when (1) every image pixel is owned by exactly one tile, (2) every
// owned pixel's dependency cone of `halo` pixels lies inside its tile or outside the image (so the tile's recomputed
// values are the exact ones), (3) the step's tile count does not exceed the launched grid; otherwise a code > 0.
extern "C" int ctrl_geometry_check(int w, int h, int tw, int th, int K, int shift, int split_warp, int *n_tiles_out,
                                   int *grid_out) {
    const Tvl1StepGeom g = tvl1_step_geom(w, h, tw, th, K, split_warp ? shift : 0);
    const int nt = g.ntx * g.nty, grid = tvl1_step_grid(w, h, tw, th, K, shift, split_warp);
    *n_tiles_out = nt;
    *grid_out = grid;
    if (nt > grid || nt < 1)
        return 3;
    if (g.halo != K)
        return 4;
    std::vector<unsigned char> owner((size_t)w * h, 0);
    for (int t = 0; t < nt; ++t) {
        const Tvl1TilePlace p = tvl1_tile_place(g, tw, th, t);
        for (int ly = 0; ly < th; ++ly)
            for (int lx = 0; lx < tw; ++lx) {
                if (!tvl1_tile_owns(g, p, tw, th, lx, ly, w, h))
                    continue;
                const int gx = p.x0 + lx, gy = p.y0 + ly;
                if (owner[(size_t)gy * w + gx]++)
                    return 1; // owned twice
                const int xa = gx - g.halo < 0 ? 0 : gx - g.halo, xb = gx + g.halo > w - 1 ? w - 1 : gx + g.halo;
                const int ya = gy - g.halo < 0 ? 0 : gy - g.halo, yb = gy + g.halo > h - 1 ? h - 1 : gy + g.halo;
                if (xa < p.x0 || xb >= p.x0 + tw || ya < p.y0 || yb >= p.y0 + th)
                    return 2; // would depend on a value the tile cannot recompute exactly
            }
    }
    for (size_t i = 0; i < owner.size(); ++i)
        if (owner[i] != 1)
            return 5; // not owned at all
    return 0;
}

// dfx_xcd_tile_index (dfx_device.h, the workgroup -> tile mapping of every kernel that reads a neighbourhood): a bijection
// of [0, nt), and the workgroups of one dispatch class (lin % 8) get one contiguous run of tiles.
#include "../denseflow_amd/csrc/dfx_device.h"
extern "C" int ctrl_xcd_map_check(int nt) {
    std::vector<int> seen((size_t)nt, 0);
    for (int lin = 0; lin < nt; ++lin) {
        const int t = dfx_xcd_tile_index(lin, nt);
        if (t < 0 || t >= nt || seen[(size_t)t]++)
            return 1;
        if (lin >= 8 && t != dfx_xcd_tile_index(lin - 8, nt) + 1)
            return 2; // the next workgroup of the same class takes the next tile
    }
    return 0;
}

// tvl1_step_work (bench.py's useful_frac): exported for the row-by-row recount of tests/test_ctrl_logic.py
extern "C" int ctrl_step_work(int n, int K) { return tvl1_step_work(n, K); }
