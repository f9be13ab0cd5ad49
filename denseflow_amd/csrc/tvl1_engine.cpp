// tvl1_engine.cpp — host control of the -a=tvl1 path (replaces cv::cuda::OpticalFlowDual_TVL1::calc
// as called at /root/reference/src/denseflow_gpu.cpp:327; algorithm: SURVEY.md Appendix A).
//
//   * all device memory is allocated once per handle (the reference re-creates the OpenCV algorithm
//     object, and with it every GpuMat, per FlowBuffer: src/denseflow_gpu.cpp:299, :345-355);
//   * a frame's float pyramid + centred gradient is built once and used as I1 of pair i and as I0
//     of pair i+step (the reference re-uploads and re-converts both frames of every pair, :317-318);
//   * `batch` pairs advance together through every launch (grid.z = pair), each with its own
//     device-side convergence state (tvl1_ctrl.h), so there is no host sync inside a pair — the
//     reference syncs the stream at every convergence check.  The host enqueues
//     groups of identical step launches and only looks at one pinned word per group to learn that
//     every pair has finished the level.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "dfx_internal.h"
#include "tvl1_kernels.h"

namespace {

struct Level {
    int w, h, pitch;
    long long off; // element offset inside a frame slot
};

class Tvl1Engine final : public AlgoEngine {
  public:
    explicit Tvl1Engine(dfx_context *ctx) : c(ctx) {}
    ~Tvl1Engine() override { destroy(); }

    int create() override;
    int batch() const override { return B; }
    int ensure_frame_slots(int need) override;
    int frame_slots() const override { return n_frame_slots; }
    int build_frames(const unsigned char *d_src, long long src_frame_stride, long long src_pitch, int n,
                     const int *h_slots) override;
    int run_pairs(int nb, const PairDesc *h_pairs, float *d_out, long long out_stride) override;
    int account(int nb) override;

  private:
    void destroy();
    Tvl1LevelCtx level_ctx(int s, int n_pairs) const;
    int steps_per_group(int s, int nb) const;

    dfx_context *c;
    int nlevels = 0;
    Level lv[DFX_LVL_MAX];
    long long frame_elems = 0;

    int n_frame_slots = 0;
    float *dI = nullptr, *dIx = nullptr, *dIy = nullptr;
    int *d_frame_slots = nullptr;
    int *h_slots_pinned = nullptr;

    int B = 0;
    float *d_planes = nullptr;
    long long plane_stride = 0, slot_stride = 0;
    Tvl1State *d_state = nullptr;
    PairDesc *d_pairs = nullptr;
    PairDesc *h_pairs_pinned = nullptr;
    double *d_partials = nullptr;
    int partials_stride = 0;
    int *d_iters_out = nullptr, *d_checks_out = nullptr;
    int *h_iters = nullptr, *h_checks = nullptr;
    long long *d_work_out = nullptr, *h_work = nullptr;
    unsigned int *d_level_done = nullptr;
    int *h_done_flag = nullptr, *d_done_flag = nullptr;
    hipEvent_t ev_group[2] = {nullptr, nullptr};
    hipEvent_t ev_lvl[DFX_LVL_MAX][2] = {};
    int done_token = 0;
    int group_override = 0;
    int min_group = 2;
    bool split_warp = false; // backward warp as its own kernel in front of every step (packed step kernels only)
    int geom = 0;            // 1 = tile columns of the step kernel start at x = 0 (Tvl1LevelCtx::geom)
    bool warp_head = false;  // the warp kernel also runs the head of the loop it starts (k_tvl1_warp_head)
    int launched_steps[DFX_LVL_MAX] = {0};

    Tvl1LoopCfg loop{};
    Tvl1Consts kc{};
};

void Tvl1Engine::destroy() {
    dfx_free_dev(dI);
    dfx_free_dev(dIx);
    dfx_free_dev(dIy);
    dfx_free_dev(d_frame_slots);
    dfx_free_host(h_slots_pinned);
    dfx_free_dev(d_planes);
    dfx_free_dev(d_state);
    dfx_free_dev(d_pairs);
    dfx_free_host(h_pairs_pinned);
    dfx_free_dev(d_partials);
    dfx_free_dev(d_iters_out);
    dfx_free_dev(d_checks_out);
    dfx_free_dev(d_work_out);
    dfx_free_host(h_work);
    dfx_free_host(h_iters);
    dfx_free_host(h_checks);
    dfx_free_dev(d_level_done);
    dfx_free_host(h_done_flag);
    for (auto &e : ev_group)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
    for (auto &e : ev_lvl)
        for (auto &x : e)
            if (x) {
                (void)hipEventDestroy(x);
                x = nullptr;
            }
}

int Tvl1Engine::create() {
    const dfx_params &p = c->prm;
    if (p.tvl1_nscales < 1 || p.tvl1_nscales > DFX_LVL_MAX || p.tvl1_warps < 0 || p.tvl1_warps > TVL1_MAX_WARPS ||
        p.tvl1_iterations < 0 || !(p.tvl1_scale_step > 0.0 && p.tvl1_scale_step < 1.0) || !(p.tvl1_theta > 0.0))
        return dfx_fail(c, DFX_ERR_INVALID, "invalid TVL1 parameters");
    if (p.impl < 0 || p.impl > 2)
        return dfx_fail(c, DFX_ERR_INVALID, "tvl1: impl must be 0 (tuned), 1 (simple) or 2 (scalar tile function)");
    if (p.tvl1_math < 0 || p.tvl1_math > 3 || (p.tvl1_math == 1 && p.impl != 0))
        return dfx_fail(c, DFX_ERR_INVALID,
                        "tvl1_math must be 0 (exact), 1 (fast; tuned kernel only), 2 (exact, sqrtf hypot) or 3 (exact, "
                        "libm hypot)");
    group_override = std::max(0, std::min(p.step_group, 64));
    // the dedicated warp kernel does not write the grad plane: only the packed tile function (impl 0) rebuilds it;
    // zero iterations: warps inside the step kernel
    split_warp = p.impl == 0 && p.tvl1_iterations > 0 && !(p.variant & DFX_VAR_TVL1_WARP_IN_STEP);
    // tile columns from x = 0 (tvl1_ctrl.h) need every warp outside the step kernel (its warp phase tiles classically)
    geom = (p.impl == 0 && split_warp && !(p.variant & DFX_VAR_TVL1_CLASSIC_GEOM)) ? 1 : 0;
    // warp + head of the loop in one launch (round 6): the tuned forms only — every cross-check variant keeps its own kernels
    warp_head = split_warp && geom == 1 && !(p.variant & (DFX_VAR_TVL1_NO_HEAD | DFX_VAR_TVL1_WARP_GATHER));

    // pyramid (A.2 step 3): cvRound(size*scaleStep) per level; a level below 16 px is discarded
    {
        long long off = 0;
        int w = c->W, h = c->H;
        nlevels = 0;
        for (int s = 0; s < p.tvl1_nscales && s < DFX_LVL_MAX; ++s) {
            if (s > 0) {
                w = dfx_cv_round(lv[s - 1].w * p.tvl1_scale_step);
                h = dfx_cv_round(lv[s - 1].h * p.tvl1_scale_step);
                if (w < 16 || h < 16)
                    break;
            }
            lv[s] = Level{w, h, dfx_round_up(w, 64), off};
            off += (long long)lv[s].pitch * h;
            nlevels = s + 1;
        }
        frame_elems = off;
    }

    loop.warps = p.tvl1_warps;
    loop.iterations = p.tvl1_iterations;
    if (p.impl == 1)
        loop.fuse_k = 1;
    else
        loop.fuse_k = std::max(1, std::min(p.tvl1_fuse_k > 0 ? p.tvl1_fuse_k : 4, tvl1_fused_max_k()));
    kc.l_t = (float)(p.tvl1_lambda * p.tvl1_theta);
    kc.taut = (float)(p.tvl1_tau / p.tvl1_theta);
    kc.theta = (float)p.tvl1_theta;
    kc.hyp = p.tvl1_math == 1 ? 0 : p.tvl1_math; // tvl1_math.h: TVL1_HYP_* (the fast mode never reaches a scalar form)

    // batch: enough pairs that the coarse levels fill 256 CUs, bounded by memory
    const long long plane = (long long)lv[0].pitch * c->H;
    plane_stride = plane;
    slot_stride = plane_stride * PL_COUNT;
    // the tile kernels address a pair slot with 32-bit byte offsets behind a buffer descriptor (tvl1_device_common.h)
    if ((unsigned long long)slot_stride * sizeof(float) >= (1ull << 32))
        return dfx_fail(c, DFX_ERR_INVALID, "tvl1: frame too large (a pair's 16 work planes must stay below 4 GB: about 8192 x 8192)");
    B = p.max_batch;
    if (B <= 0) {
        const long long px0 = (long long)c->W * c->H;
        B = (int)std::max<long long>(1, std::min<long long>(DFX_MAX_BATCH, (256LL << 20) / std::max<long long>(px0, 1)));
    }
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    const size_t per_pair = (size_t)slot_stride * 4 + (size_t)c->W * c->H * 9 + (size_t)frame_elems * 12;
    while (B > 1 && per_pair * (size_t)(B + 2) > free_b / 2)
        B /= 2;

    HIPCHK(c, hipMalloc(&d_planes, (size_t)slot_stride * B * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_state, sizeof(Tvl1State) * B));
    HIPCHK(c, hipMemset(d_state, 0, sizeof(Tvl1State) * B));
    HIPCHK(c, hipMalloc(&d_pairs, sizeof(PairDesc) * B));
    HIPCHK(c, hipHostMalloc(&h_pairs_pinned, sizeof(PairDesc) * B, hipHostMallocDefault));
    partials_stride = ((lv[0].w + 63) / 64) * ((c->H + 3) / 4) + 64; // >= workgroups of any step variant
    HIPCHK(c, hipMalloc(&d_partials, sizeof(double) * (size_t)partials_stride * B));
    HIPCHK(c, hipMalloc(&d_iters_out, sizeof(int) * B * DFX_LVL_MAX * TVL1_MAX_WARPS));
    HIPCHK(c, hipMalloc(&d_checks_out, sizeof(int) * B * DFX_LVL_MAX * 2));
    HIPCHK(c, hipMalloc(&d_work_out, sizeof(long long) * B * DFX_LVL_MAX * 2));
    HIPCHK(c, hipMemset(d_work_out, 0, sizeof(long long) * B * DFX_LVL_MAX * 2));
    HIPCHK(c, hipHostMalloc(&h_work, sizeof(long long) * B * DFX_LVL_MAX * 2, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc(&h_iters, sizeof(int) * B * DFX_LVL_MAX * TVL1_MAX_WARPS, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc(&h_checks, sizeof(int) * B * DFX_LVL_MAX * 2, hipHostMallocDefault));
    HIPCHK(c, hipMalloc(&d_level_done, sizeof(unsigned int)));
    HIPCHK(c, hipMemset(d_level_done, 0, sizeof(unsigned int)));
    HIPCHK(c, hipHostMalloc(&h_done_flag, 64, hipHostMallocMapped));
    *h_done_flag = 0;
    HIPCHK(c, hipHostGetDevicePointer((void **)&d_done_flag, h_done_flag, 0));
    HIPCHK(c, hipEventCreateWithFlags(&ev_group[0], dfx_event_flags(c, false)));
    HIPCHK(c, hipEventCreateWithFlags(&ev_group[1], dfx_event_flags(c, false)));
    for (auto &e : ev_lvl) {
        HIPCHK(c, hipEventCreateWithFlags(&e[0], dfx_event_flags(c, true)));
        HIPCHK(c, hipEventCreateWithFlags(&e[1], dfx_event_flags(c, true)));
    }
    return ensure_frame_slots(B + 1);
}

int Tvl1Engine::ensure_frame_slots(int need) {
    if (need <= n_frame_slots)
        return DFX_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfx_free_dev(dI);
    dfx_free_dev(dIx);
    dfx_free_dev(dIy);
    dfx_free_dev(d_frame_slots);
    dfx_free_host(h_slots_pinned);
    const size_t bytes = (size_t)need * frame_elems * sizeof(float);
    HIPCHK(c, hipMalloc(&dI, bytes));
    HIPCHK(c, hipMalloc(&dIx, bytes));
    HIPCHK(c, hipMalloc(&dIy, bytes));
    HIPCHK(c, hipMalloc(&d_frame_slots, sizeof(int) * need));
    HIPCHK(c, hipHostMalloc(&h_slots_pinned, sizeof(int) * need, hipHostMallocDefault));
    n_frame_slots = need;
    return DFX_OK;
}

// float pyramids + centred gradients (A.2 steps 1-3, A.3) of `n` new frames
int Tvl1Engine::build_frames(const unsigned char *d_src, long long src_frame_stride, long long src_pitch, int n,
                             const int *h_slots) {
    if (n <= 0)
        return DFX_OK;
    std::memcpy(h_slots_pinned, h_slots, sizeof(int) * n);
    HIPCHK(c, hipMemcpyAsync(d_frame_slots, h_slots_pinned, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    const Level &L0 = lv[0];
    tvl1_launch_u8_to_f32(c->stream, d_src, src_frame_stride, src_pitch, d_frame_slots, n, dI, frame_elems, L0.w, L0.h,
                          L0.pitch);
    const float ifs = (float)(1.0 / c->prm.tvl1_scale_step); // the given fx is kept (E.1)
    for (int s = 1; s < nlevels; ++s) {
        const Level &A = lv[s - 1], &Bq = lv[s];
        tvl1_launch_pyr_down(c->stream, dI, frame_elems, d_frame_slots, n, A.off, A.w, A.h, A.pitch, Bq.off, Bq.w, Bq.h,
                             Bq.pitch, ifs, ifs);
    }
    for (int s = 0; s < nlevels; ++s) {
        const Level &L = lv[s];
        tvl1_launch_centered_gradient(c->stream, dI, dIx, dIy, frame_elems, d_frame_slots, n, L.off, L.w, L.h, L.pitch);
    }
    c->stats.kernel_launches += 1 + (nlevels - 1) + nlevels;
    return DFX_OK;
}

Tvl1LevelCtx Tvl1Engine::level_ctx(int s, int n_pairs) const {
    Tvl1LevelCtx x;
    std::memset(&x, 0, sizeof x);
    const Level &L = lv[s];
    x.w = L.w;
    x.h = L.h;
    x.pitch = L.pitch;
    x.frame_I = dI;
    x.frame_Ix = dIx;
    x.frame_Iy = dIy;
    x.frame_stride = frame_elems;
    x.lvl_off = L.off;
    x.planes = d_planes;
    x.plane_stride = plane_stride;
    x.slot_stride = slot_stride;
    x.state = d_state;
    x.pairs = d_pairs;
    x.partials = d_partials;
    x.partials_stride = partials_stride;
    x.n_pairs = n_pairs;
    x.loop = loop;
    x.k = kc;
    x.thr = c->prm.tvl1_epsilon * c->prm.tvl1_epsilon * (double)(L.w * L.h);
    x.level = s;
    x.iters_out = d_iters_out;
    x.checks_out = d_checks_out;
    x.work_out = d_work_out;
    x.level_done_count = d_level_done;
    x.host_done_flag = d_done_flag;
    x.done_token = 0;
    x.split_warp = split_warp ? 1 : 0;
    x.warp_lds = (c->prm.variant & DFX_VAR_TVL1_WARP_GATHER) ? 0 : 1;
    x.geom = geom;
    x.head = warp_head ? 1 : 0;
    return x;
}

int Tvl1Engine::steps_per_group(int s, int nb) const {
    if (group_override > 0)
        return group_override;
    // aim at >= ~150 us of device work per group: the host stays ahead of the device and the event
    // record between groups (~6 us of idle queue) stays below a few percent
    const double px = (double)lv[s].w * lv[s].h * nb;
    const double step_us = 2.0 + px * 64.0 * loop.fuse_k / 4.0e6; // bytes / (4 TB/s) in us
    const int g = (int)std::ceil(150.0 / step_us);
    // at least 2: the launches enqueued behind a level's last useful step (up to two groups: the host looks at the group
    // before the one it has just enqueued) find nothing to do and cost 14-57 us each at 1080p x 129 pairs
    return std::max(min_group, std::min(16, g));
}

int Tvl1Engine::run_pairs(int nb, const PairDesc *h_pairs, float *d_out, long long out_stride) {
    std::memcpy(h_pairs_pinned, h_pairs, sizeof(PairDesc) * nb);
    HIPCHK(c, hipMemcpyAsync(d_pairs, h_pairs_pinned, sizeof(PairDesc) * nb, hipMemcpyHostToDevice, c->stream));
    const int impl = c->prm.impl, math = c->prm.tvl1_math;
    const float up = (float)(1.0 / c->prm.tvl1_scale_step);
    const int hard_limit = loop.warps * (loop.iterations + 2) + 64;

    for (int s = nlevels - 1; s >= 0; --s) {
        Tvl1LevelCtx x = level_ctx(s, nb);
        x.done_token = ++done_token;
        tvl1_launch_level_begin(c->stream, x, s == nlevels - 1);
        c->stats.kernel_launches += (warp_head && s != nlevels - 1) ? 1 : 2;
        launched_steps[s] = 0;
        if (loop.warps > 0) {
            const int G = steps_per_group(s, nb);
            int step_id = 0;
            HIPCHK(c, hipEventRecord(ev_lvl[s][0], c->stream));
            for (int g = 0;; ++g) {
                for (int i = 0; i < G; ++i) {
                    if (warp_head)
                        tvl1_launch_warp_head(c->stream, x, step_id, math);
                    else if (split_warp)
                        tvl1_launch_warp(c->stream, x, step_id);
                    tvl1_launch_step(c->stream, x, step_id++, impl, math);
                }
                c->stats.kernel_launches += (uint64_t)G * (split_warp ? 2 : 1);
                HIPCHK(c, hipEventRecord(ev_group[g & 1], c->stream));
                if (g >= 1) { // look at the group before the one just enqueued: the device never idles
                    HIPCHK(c, hipEventSynchronize(ev_group[(g - 1) & 1]));
                    if (*(volatile int *)h_done_flag == x.done_token)
                        break;
                }
                if (step_id > hard_limit + 2 * G) {
                    HIPCHK(c, dfx_stream_wait(c, c->stream));
                    if (*(volatile int *)h_done_flag == x.done_token)
                        break;
                    return dfx_fail(c, DFX_ERR_HIP, "TVL1 level did not terminate within its step bound");
                }
            }
            launched_steps[s] = step_id;
            HIPCHK(c, hipEventRecord(ev_lvl[s][1], c->stream));
        }
        if (s > 0) {
            const Level &D = lv[s - 1], &S = lv[s];
            const float ifx = (float)(1.0 / ((double)D.w / (double)S.w));
            const float ify = (float)(1.0 / ((double)D.h / (double)S.h));
            tvl1_launch_upsample_u(c->stream, x, D.w, D.h, D.pitch, ifx, ify, up);
        } else {
            tvl1_launch_merge(c->stream, x, d_out, out_stride);
        }
        c->stats.kernel_launches += 1;
    }
    HIPCHK(c, hipMemcpyAsync(h_iters, d_iters_out, sizeof(int) * nb * DFX_LVL_MAX * TVL1_MAX_WARPS,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_checks, d_checks_out, sizeof(int) * nb * DFX_LVL_MAX * 2, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(h_work, d_work_out, sizeof(long long) * nb * DFX_LVL_MAX * 2, hipMemcpyDeviceToHost,
                             c->stream));
    return DFX_OK;
}

// Fold the read-back iteration counts of a finished batch into the statistics (SURVEY.md §8d byte model).
int Tvl1Engine::account(int nb) {
    dfx_stats &st = c->stats;
    for (int s = 0; s < nlevels && loop.warps > 0; ++s) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, ev_lvl[s][0], ev_lvl[s][1]));
        st.step_ms += ms;
        st.step_launches += (uint64_t)launched_steps[s];
        st.level_ms[s] += ms;
        st.level_launches[s] += (uint64_t)launched_steps[s];
        int useful = 0;
        for (int b = 0; b < nb; ++b)
            useful = std::max(useful, h_checks[(b * DFX_LVL_MAX + s) * 2 + 1]);
        st.noop_steps += (uint64_t)std::max(0, launched_steps[s] - useful);
    }
    int step_tiles[DFX_LVL_MAX] = {0}, head_tiles[DFX_LVL_MAX] = {0}; // workgroups per pair of the two launches of a step
    for (int s = 0; s < nlevels && c->prm.impl == 0; ++s) {
        const Tvl1LevelCtx x = level_ctx(s, 1);
        step_tiles[s] = tvl1_step_blocks(x, 0);
        head_tiles[s] = tvl1_head_blocks(x);
    }
    for (int b = 0; b < nb; ++b) {
        for (int s = 0; s < nlevels; ++s) {
            const double px = (double)lv[s].w * lv[s].h;
            long long it = 0;
            for (int w = 0; w < loop.warps; ++w)
                it += h_iters[(b * DFX_LVL_MAX + s) * TVL1_MAX_WARPS + w];
            st.tvl1_total_iters += (uint64_t)it;
            st.tvl1_px_iters += px * (double)it;
            // lane-iterations the tuned kernels executed: half rows x 32 lanes x tiles (tvl1_ctrl.h: tvl1_step_work)
            st.tvl1_lane_iters += 32.0 * ((double)h_work[(b * DFX_LVL_MAX + s) * 2 + 0] * step_tiles[s] +
                                          (double)h_work[(b * DFX_LVL_MAX + s) * 2 + 1] * head_tiles[s]);
            st.algorithmic_bytes += px * (64.0 * (double)it + 44.0 * loop.warps + 28.0);
            st.step_algorithmic_bytes += px * (64.0 * (double)it + 44.0 * loop.warps);
        }
        st.pairs += 1;
    }
    const int b = nb - 1;
    st.levels = nlevels;
    st.tvl1_checks = 0;
    for (int s = 0; s < nlevels; ++s) {
        st.level_w[s] = lv[s].w;
        st.level_h[s] = lv[s].h;
        for (int w = 0; w < DFX_MAX_WARPS; ++w)
            st.tvl1_iters[s][w] = (w < TVL1_MAX_WARPS) ? h_iters[(b * DFX_LVL_MAX + s) * TVL1_MAX_WARPS + w] : 0;
        st.tvl1_checks += h_checks[(b * DFX_LVL_MAX + s) * 2];
    }
    return DFX_OK;
}

} // namespace

AlgoEngine *dfx_make_tvl1_engine(dfx_context *c) { return new Tvl1Engine(c); }
