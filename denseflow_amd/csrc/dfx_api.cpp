// dfx_api.cpp — implementation of the C ABI in include/dfx.h (host control code, HIP runtime).
//
// Reference behaviour mirrored here: DenseFlow::calc_optflows_imp, src/denseflow_gpu.cpp:282-370
// (pair selection :315-316, per-pair upload/calc/download :317-339).  What is different by design:
//   * all device memory is owned by the handle and allocated once (the reference re-creates the
//     OpenCV algorithm object, and with it every GpuMat, per FlowBuffer: :299-303, :345-355);
//   * a frame's float pyramid + gradient is built once and reused as I1 of pair i and I0 of pair
//     i+step (the reference re-uploads and re-converts it: :317-318);
//   * `max_batch` pairs advance together through every launch (grid.z = pair), each with its own
//     device-side convergence state, so there is no host sync inside a pair (the reference syncs
//     the stream at every convergence check).
#include "../../include/dfx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dfx_device.h"
#include "tvl1_kernels.h"

namespace {

thread_local std::string g_create_error;

inline int cv_round(double v) { return (int)std::lrint(v); } // round-half-even

struct Level {
    int w, h, pitch;
    long long off; // element offset inside a frame slot
};

} // namespace

struct dfx_context {
    int device = 0;
    dfx_algo algo = DFX_ALGO_TVL1;
    int W = 0, H = 0;
    dfx_params prm{};
    std::string err;

    hipStream_t stream = nullptr;
    hipEvent_t ev_group[2] = {nullptr, nullptr};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    hipEvent_t ev_lvl[DFX_LVL_MAX][2] = {};

    // pyramid geometry
    int nlevels = 0;
    Level lv[DFX_LVL_MAX];
    long long frame_elems = 0; // elements per pyramid per frame slot

    // frame cache
    int n_frame_slots = 0;
    float *dI = nullptr, *dIx = nullptr, *dIy = nullptr;
    unsigned char *d_u8 = nullptr; // staging for host frames: n_frame_slots dense W*H images
    std::vector<long long> slot_holds;
    long long frames_built = 0; // frame ids [0, frames_built) of the current call have pyramids

    // pair slots
    int B = 0;
    float *d_planes = nullptr;
    long long plane_stride = 0, slot_stride = 0;
    Tvl1State *d_state = nullptr;
    PairDesc *d_pairs = nullptr;
    int *d_frame_slots = nullptr;
    double *d_partials = nullptr;
    int partials_stride = 0;
    int *d_iters_out = nullptr;
    int *d_checks_out = nullptr;
    unsigned int *d_level_done = nullptr;
    int *h_done_flag = nullptr; // pinned, mapped
    int *d_done_flag = nullptr; // device view of h_done_flag
    PairDesc *h_pairs = nullptr; // pinned staging
    int *h_slots = nullptr;      // pinned staging
    int *h_iters = nullptr;      // pinned readback [B][DFX_LVL_MAX][TVL1_MAX_WARPS]
    int *h_checks = nullptr;     // pinned readback [B][DFX_LVL_MAX][2]
    float *d_flow_out = nullptr; // [B][H*W*2] when the destination is host memory
    int done_token = 0;
    int group_override = 0;
    int launched_steps[DFX_LVL_MAX] = {0};

    Tvl1LoopCfg loop{};
    Tvl1Consts kc{};

    dfx_stats stats{};
};

namespace {

#define HIPCHK(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            char buf_[512];                                                                                     \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,        \
                     __LINE__);                                                                                 \
            (ctx)->err = buf_;                                                                                  \
            return DFX_ERR_HIP;                                                                                 \
        }                                                                                                       \
    } while (0)

int fail(dfx_context *c, int code, const std::string &msg) {
    if (c)
        c->err = msg;
    else
        g_create_error = msg;
    return code;
}

void default_params(dfx_params *p) {
    std::memset(p, 0, sizeof *p);
    // cv::cuda::OpticalFlowDual_TVL1::create() defaults
    p->tvl1_tau = 0.25;
    p->tvl1_lambda = 0.15;
    p->tvl1_theta = 0.3;
    p->tvl1_nscales = 5;
    p->tvl1_warps = 5;
    p->tvl1_epsilon = 0.01;
    p->tvl1_iterations = 300;
    p->tvl1_scale_step = 0.8;
    // cv::cuda::FarnebackOpticalFlow::create() defaults
    p->farn_num_levels = 5;
    p->farn_pyr_scale = 0.5;
    p->farn_win_size = 13;
    p->farn_num_iters = 10;
    p->farn_poly_n = 5;
    p->farn_poly_sigma = 1.1;
    p->farn_flags = 0;
    // cuda::BroxOpticalFlow::create(0.197f, 50.0f, 0.8f, 10, 77, 10): src/denseflow_gpu.cpp:303
    p->brox_alpha = 0.197f;
    p->brox_gamma = 50.0f;
    p->brox_scale_factor = 0.8f;
    p->brox_inner_iterations = 10;
    p->brox_outer_iterations = 77;
    p->brox_solver_iterations = 10;
}

// TVL1 pyramid (A.2 step 3): cvRound(size*scaleStep) per level, discarded below 16 px.
int build_tvl1_levels(dfx_context *c) {
    const dfx_params &p = c->prm;
    int n = 0;
    long long off = 0;
    int w = c->W, h = c->H;
    for (int s = 0; s < p.tvl1_nscales && s < DFX_LVL_MAX; ++s) {
        if (s > 0) {
            w = cv_round(c->lv[s - 1].w * p.tvl1_scale_step);
            h = cv_round(c->lv[s - 1].h * p.tvl1_scale_step);
            if (w < 16 || h < 16)
                break;
        }
        Level &L = c->lv[s];
        L.w = w;
        L.h = h;
        L.pitch = dfx_round_up(w, 64);
        L.off = off;
        off += (long long)L.pitch * h;
        n = s + 1;
    }
    c->nlevels = n;
    c->frame_elems = off;
    return n;
}

void destroy_buffers(dfx_context *c) {
    auto F = [](auto *&p) {
        if (p) {
            (void)hipFree(p);
            p = nullptr;
        }
    };
    auto FH = [](auto *&p) {
        if (p) {
            (void)hipHostFree(p);
            p = nullptr;
        }
    };
    F(c->dI);
    F(c->dIx);
    F(c->dIy);
    F(c->d_u8);
    F(c->d_planes);
    F(c->d_state);
    F(c->d_pairs);
    F(c->d_frame_slots);
    F(c->d_partials);
    F(c->d_iters_out);
    F(c->d_checks_out);
    F(c->d_level_done);
    F(c->d_flow_out);
    FH(c->h_done_flag);
    FH(c->h_pairs);
    FH(c->h_slots);
    FH(c->h_iters);
    FH(c->h_checks);
}

int ensure_frame_slots(dfx_context *c, int need) {
    if (need <= c->n_frame_slots)
        return DFX_OK;
    if (c->dI) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->dI);
        (void)hipFree(c->dIx);
        (void)hipFree(c->dIy);
        (void)hipFree(c->d_u8);
        (void)hipFree(c->d_frame_slots);
        (void)hipHostFree(c->h_slots);
        c->dI = c->dIx = c->dIy = nullptr;
        c->d_u8 = nullptr;
        c->d_frame_slots = nullptr;
        c->h_slots = nullptr;
    }
    const size_t bytes = (size_t)need * c->frame_elems * sizeof(float);
    HIPCHK(c, hipMalloc(&c->dI, bytes));
    HIPCHK(c, hipMalloc(&c->dIx, bytes));
    HIPCHK(c, hipMalloc(&c->dIy, bytes));
    HIPCHK(c, hipMalloc(&c->d_u8, (size_t)need * c->W * c->H));
    HIPCHK(c, hipMalloc(&c->d_frame_slots, sizeof(int) * need));
    HIPCHK(c, hipHostMalloc(&c->h_slots, sizeof(int) * need, hipHostMallocDefault));
    c->n_frame_slots = need;
    c->slot_holds.assign(need, -1);
    c->frames_built = 0;
    return DFX_OK;
}

Tvl1LevelCtx make_level_ctx(dfx_context *c, int s, int n_pairs) {
    Tvl1LevelCtx x;
    std::memset(&x, 0, sizeof x);
    const Level &L = c->lv[s];
    x.w = L.w;
    x.h = L.h;
    x.pitch = L.pitch;
    x.frame_I = c->dI;
    x.frame_Ix = c->dIx;
    x.frame_Iy = c->dIy;
    x.frame_stride = c->frame_elems;
    x.lvl_off = L.off;
    x.planes = c->d_planes;
    x.plane_stride = c->plane_stride;
    x.slot_stride = c->slot_stride;
    x.state = c->d_state;
    x.pairs = c->d_pairs;
    x.partials = c->d_partials;
    x.partials_stride = c->partials_stride;
    x.n_pairs = n_pairs;
    x.loop = c->loop;
    x.k = c->kc;
    x.thr = c->prm.tvl1_epsilon * c->prm.tvl1_epsilon * (double)(L.w * L.h);
    x.level = s;
    x.iters_out = c->d_iters_out;
    x.checks_out = c->d_checks_out;
    x.level_done_count = c->d_level_done;
    x.host_done_flag = c->d_done_flag;
    x.done_token = 0;
    return x;
}

int create_tvl1(dfx_context *c) {
    const dfx_params &p = c->prm;
    if (p.tvl1_nscales < 1 || p.tvl1_nscales > DFX_LVL_MAX || p.tvl1_warps < 0 || p.tvl1_warps > TVL1_MAX_WARPS ||
        p.tvl1_iterations < 0 || !(p.tvl1_scale_step > 0.0 && p.tvl1_scale_step < 1.0) || !(p.tvl1_theta > 0.0))
        return fail(c, DFX_ERR_INVALID, "invalid TVL1 parameters");
    build_tvl1_levels(c);

    c->loop.warps = p.tvl1_warps;
    c->loop.iterations = p.tvl1_iterations;
    if (p.impl == 1)
        c->loop.fuse_k = 1;
    else
        c->loop.fuse_k = std::max(1, std::min(p.tvl1_fuse_k > 0 ? p.tvl1_fuse_k : 4, tvl1_fused_max_k(p.tvl1_tile_h)));
    c->kc.l_t = (float)(p.tvl1_lambda * p.tvl1_theta);
    c->kc.taut = (float)(p.tvl1_tau / p.tvl1_theta);
    c->kc.theta = (float)p.tvl1_theta;

    // batch size: enough pairs to give every CU several workgroups at the coarsest level, bounded by memory
    int B = p.max_batch;
    if (B <= 0) {
        const long long px0 = (long long)c->W * c->H;
        B = (int)std::max<long long>(1, std::min<long long>(64, (4LL << 20) / std::max<long long>(px0, 1)));
    }
    const long long plane = (long long)c->lv[0].pitch * c->H;
    c->plane_stride = plane;
    c->slot_stride = plane * PL_COUNT;
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    const size_t per_pair = (size_t)c->slot_stride * 4 + (size_t)c->W * c->H * 8 + (size_t)c->frame_elems * 12;
    while (B > 1 && per_pair * (size_t)(B + 2) > free_b / 2)
        B /= 2;
    c->B = B;

    HIPCHK(c, hipMalloc(&c->d_planes, (size_t)c->slot_stride * B * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->d_state, sizeof(Tvl1State) * B));
    HIPCHK(c, hipMemset(c->d_state, 0, sizeof(Tvl1State) * B));
    HIPCHK(c, hipMalloc(&c->d_pairs, sizeof(PairDesc) * B));
    c->partials_stride = ((c->lv[0].w + 63) / 64) * ((c->H + 3) / 4) + 64;
    HIPCHK(c, hipMalloc(&c->d_partials, sizeof(double) * (size_t)c->partials_stride * B));
    HIPCHK(c, hipMalloc(&c->d_iters_out, sizeof(int) * B * DFX_LVL_MAX * TVL1_MAX_WARPS));
    HIPCHK(c, hipMalloc(&c->d_checks_out, sizeof(int) * B * DFX_LVL_MAX * 2));
    HIPCHK(c, hipMalloc(&c->d_level_done, sizeof(unsigned int)));
    HIPCHK(c, hipMemset(c->d_level_done, 0, sizeof(unsigned int)));
    HIPCHK(c, hipHostMalloc(&c->h_done_flag, 64, hipHostMallocMapped));
    *c->h_done_flag = 0;
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->d_done_flag, c->h_done_flag, 0));
    HIPCHK(c, hipHostMalloc(&c->h_pairs, sizeof(PairDesc) * B, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc(&c->h_iters, sizeof(int) * B * DFX_LVL_MAX * TVL1_MAX_WARPS, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc(&c->h_checks, sizeof(int) * B * DFX_LVL_MAX * 2, hipHostMallocDefault));
    HIPCHK(c, hipMalloc(&c->d_flow_out, (size_t)B * c->W * c->H * 2 * sizeof(float)));
    return ensure_frame_slots(c, B + 1);
}

// Build float pyramids + centred gradients for `n` frames whose u8 pixels start at d_src
// (frame z at d_src + z*src_frame_stride); slot ids are in c->h_slots[0..n).
int build_frames(dfx_context *c, const unsigned char *d_src, long long src_frame_stride, long long src_pitch, int n) {
    if (n <= 0)
        return DFX_OK;
    HIPCHK(c, hipMemcpyAsync(c->d_frame_slots, c->h_slots, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    const Level &L0 = c->lv[0];
    tvl1_launch_u8_to_f32(c->stream, d_src, src_frame_stride, src_pitch, c->d_frame_slots, n, c->dI, c->frame_elems,
                          L0.w, L0.h, L0.pitch);
    const float ifs = (float)(1.0 / c->prm.tvl1_scale_step); // the given fx is kept (E.1)
    for (int s = 1; s < c->nlevels; ++s) {
        const Level &A = c->lv[s - 1], &Bq = c->lv[s];
        tvl1_launch_pyr_down(c->stream, c->dI, c->frame_elems, c->d_frame_slots, n, A.off, A.w, A.h, A.pitch, Bq.off,
                             Bq.w, Bq.h, Bq.pitch, ifs, ifs);
    }
    for (int s = 0; s < c->nlevels; ++s) {
        const Level &L = c->lv[s];
        tvl1_launch_centered_gradient(c->stream, c->dI, c->dIx, c->dIy, c->frame_elems, c->d_frame_slots, n, L.off,
                                      L.w, L.h, L.pitch);
    }
    c->stats.kernel_launches += 1 + (c->nlevels - 1) + c->nlevels;
    return DFX_OK;
}

int steps_per_group(const dfx_context *c, int s, int nb) {
    if (c->group_override > 0)
        return c->group_override;
    // aim at >= ~150 us of device work per group: the host stays ahead of the device and the event
    // record between groups (~6 us of idle queue) stays below a few percent
    const double px = (double)c->lv[s].w * c->lv[s].h * nb;
    const double step_us = 2.0 + px * 64.0 * c->loop.fuse_k / 4.0e6; // bytes / (4 TB/s) in us
    int g = (int)std::ceil(150.0 / step_us);
    return std::max(6, std::min(16, g));
}

// Run the TVL1 pyramid for `nb` pairs whose descriptors are in c->h_pairs; flows go to d_out.
int run_tvl1_pairs(dfx_context *c, int nb, float *d_out, long long out_stride) {
    HIPCHK(c, hipMemcpyAsync(c->d_pairs, c->h_pairs, sizeof(PairDesc) * nb, hipMemcpyHostToDevice, c->stream));
    const int impl = c->prm.impl;
    const float up = (float)(1.0 / c->prm.tvl1_scale_step);
    const int hard_limit = c->loop.warps * (c->loop.iterations + 2) + 64;

    for (int s = c->nlevels - 1; s >= 0; --s) {
        Tvl1LevelCtx x = make_level_ctx(c, s, nb);
        x.done_token = ++c->done_token;
        tvl1_launch_level_begin(c->stream, x, s == c->nlevels - 1);
        c->stats.kernel_launches += 2;
        if (c->loop.warps > 0) {
            const int G = steps_per_group(c, s, nb);
            int step_id = 0;
            HIPCHK(c, hipEventRecord(c->ev_lvl[s][0], c->stream));
            for (int g = 0;; ++g) {
                for (int i = 0; i < G; ++i)
                    tvl1_launch_step(c->stream, x, step_id++, impl, c->prm.tvl1_tile_h);
                c->stats.kernel_launches += G;
                HIPCHK(c, hipEventRecord(c->ev_group[g & 1], c->stream));
                if (g >= 1) {
                    HIPCHK(c, hipEventSynchronize(c->ev_group[(g - 1) & 1]));
                    if (*(volatile int *)c->h_done_flag == x.done_token)
                        break;
                }
                if (step_id > hard_limit + 2 * G) {
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    if (*(volatile int *)c->h_done_flag == x.done_token)
                        break;
                    return fail(c, DFX_ERR_HIP, "TVL1 level did not terminate within its step bound");
                }
            }
            c->launched_steps[s] = step_id;
            HIPCHK(c, hipEventRecord(c->ev_lvl[s][1], c->stream));
        }
        if (s > 0) {
            const Level &D = c->lv[s - 1], &S = c->lv[s];
            const float ifx = (float)(1.0 / ((double)D.w / (double)S.w));
            const float ify = (float)(1.0 / ((double)D.h / (double)S.h));
            tvl1_launch_upsample_u(c->stream, x, D.w, D.h, D.pitch, ifx, ify, up);
            c->stats.kernel_launches += 1;
        } else {
            tvl1_launch_merge(c->stream, x, d_out, out_stride);
            c->stats.kernel_launches += 1;
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->h_iters, c->d_iters_out, sizeof(int) * nb * DFX_LVL_MAX * TVL1_MAX_WARPS,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_checks, c->d_checks_out, sizeof(int) * nb * DFX_LVL_MAX * 2, hipMemcpyDeviceToHost,
                             c->stream));
    return DFX_OK;
}

// Fold the read-back iteration counts of a finished batch into the statistics (SURVEY.md §8d byte model).
void account_tvl1(dfx_context *c, int nb) {
    dfx_stats &st = c->stats;
    for (int s = 0; s < c->nlevels; ++s) {
        int useful = 0;
        for (int b = 0; b < nb; ++b)
            useful = std::max(useful, c->h_checks[(b * DFX_LVL_MAX + s) * 2 + 1]);
        st.noop_steps += (uint64_t)std::max(0, c->launched_steps[s] - useful);
    }
    for (int b = 0; b < nb; ++b) {
        for (int s = 0; s < c->nlevels; ++s) {
            const double px = (double)c->lv[s].w * c->lv[s].h;
            long long it = 0;
            for (int w = 0; w < c->loop.warps; ++w)
                it += c->h_iters[(b * DFX_LVL_MAX + s) * TVL1_MAX_WARPS + w];
            st.tvl1_total_iters += (uint64_t)it;
            st.tvl1_px_iters += px * (double)it;
            st.algorithmic_bytes += px * (64.0 * (double)it + 44.0 * c->loop.warps + 28.0);
        }
        st.pairs += 1;
    }
    const int b = nb - 1;
    st.levels = c->nlevels;
    st.tvl1_checks = 0;
    for (int s = 0; s < c->nlevels; ++s) {
        st.level_w[s] = c->lv[s].w;
        st.level_h[s] = c->lv[s].h;
        for (int w = 0; w < DFX_MAX_WARPS; ++w)
            st.tvl1_iters[s][w] = (w < TVL1_MAX_WARPS) ? c->h_iters[(b * DFX_LVL_MAX + s) * TVL1_MAX_WARPS + w] : 0;
        st.tvl1_checks += c->h_checks[(b * DFX_LVL_MAX + s) * 2];
    }
}

// Shared driver for host- and device-resident frames.
//   host mode  : frames[i] host pointers (frame_pitch), flows[i] host pointers (out_pitch bytes)
//   device mode: d_frames / d_flows contiguous device arrays
int calc_batch_impl(dfx_context *c, const uint8_t *const *frames, size_t frame_pitch, const uint8_t *d_frames,
                    size_t d_pitch, size_t d_frame_stride, int n_frames, int step, float *const *flows,
                    size_t out_pitch, float *d_flows, size_t d_flow_stride) {
    if (n_frames < 0 || step == 0)
        return fail(c, DFX_ERR_INVALID, "n_frames must be >= 0 and step non-zero");
    const int astep = std::abs(step);
    const int M = std::max(n_frames - astep, 0);
    if (M == 0)
        return DFX_OK;
    if (c->algo != DFX_ALGO_TVL1)
        return fail(c, DFX_ERR_UNSUPPORTED, "algorithm not implemented yet");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_frame_slots(c, c->B + astep);
    if (rc != DFX_OK)
        return rc;
    const bool host_mode = frames != nullptr;
    std::fill(c->slot_holds.begin(), c->slot_holds.end(), -1);
    c->frames_built = 0;
    const int F = c->n_frame_slots;

    for (int i0 = 0; i0 < M; i0 += c->B) {
        const int nb = std::min(c->B, M - i0);
        HIPCHK(c, hipEventRecord(c->ev_t0, c->stream));
        // frames [i0, i0+nb+astep) must be resident; ids below frames_built already are
        const long long need_end = (long long)i0 + nb + astep;
        const long long first_new = std::max<long long>(c->frames_built, i0);
        const int n_new = (int)(need_end - first_new);
        for (int k = 0; k < n_new; ++k) {
            const long long id = first_new + k;
            const int slot = (int)(id % F);
            c->h_slots[k] = slot;
            c->slot_holds[slot] = id;
        }
        if (n_new > 0) {
            if (host_mode) {
                for (int k = 0; k < n_new; ++k)
                    HIPCHK(c, hipMemcpy2DAsync(c->d_u8 + (size_t)k * c->W * c->H, c->W, frames[first_new + k],
                                               frame_pitch, c->W, c->H, hipMemcpyHostToDevice, c->stream));
                rc = build_frames(c, c->d_u8, (long long)c->W * c->H, c->W, n_new);
            } else {
                rc = build_frames(c, d_frames + (size_t)first_new * d_frame_stride, (long long)d_frame_stride,
                                  (long long)d_pitch, n_new);
            }
            if (rc != DFX_OK)
                return rc;
            c->frames_built = need_end;
        }
        // pair i: a = (step>0 ? i : i-step), b = (step>0 ? i+step : i)   (src/denseflow_gpu.cpp:315-316)
        for (int k = 0; k < nb; ++k) {
            const int i = i0 + k;
            const int a = step > 0 ? i : i - step;
            const int b = step > 0 ? i + step : i;
            c->h_pairs[k].frame_a = a % F;
            c->h_pairs[k].frame_b = b % F;
        }
        float *dst = host_mode ? c->d_flow_out : d_flows + (size_t)i0 * d_flow_stride;
        const long long dst_stride = host_mode ? (long long)c->W * c->H * 2 : (long long)d_flow_stride;
        rc = run_tvl1_pairs(c, nb, dst, dst_stride);
        if (rc != DFX_OK)
            return rc;
        HIPCHK(c, hipEventRecord(c->ev_t1, c->stream));
        if (host_mode) {
            for (int k = 0; k < nb; ++k)
                HIPCHK(c, hipMemcpy2DAsync(flows[i0 + k], out_pitch, c->d_flow_out + (size_t)k * c->W * c->H * 2,
                                           (size_t)c->W * 8, (size_t)c->W * 8, c->H, hipMemcpyDeviceToHost,
                                           c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
        c->stats.device_ms += ms;
        for (int s = 0; s < c->nlevels && c->loop.warps > 0; ++s) {
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev_lvl[s][0], c->ev_lvl[s][1]));
            c->stats.step_ms += ms;
            c->stats.step_launches += (uint64_t)c->launched_steps[s];
            c->stats.level_ms[s] += ms;
            c->stats.level_launches[s] += (uint64_t)c->launched_steps[s];
        }
        account_tvl1(c, nb);
    }
    return DFX_OK;
}

} // namespace

// ================================================================================================
// C ABI

extern "C" {

int dfx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

void dfx_default_params(dfx_params *p) {
    if (p)
        default_params(p);
}

int dfx_algo_from_name(const char *name, dfx_algo *out) {
    if (!name)
        return DFX_ERR_UNKNOWN_ALGO;
    if (!std::strcmp(name, "tvl1")) {
        if (out) *out = DFX_ALGO_TVL1;
        return DFX_OK;
    }
    if (!std::strcmp(name, "farn")) {
        if (out) *out = DFX_ALGO_FARN;
        return DFX_OK;
    }
    if (!std::strcmp(name, "brox")) {
        if (out) *out = DFX_ALGO_BROX;
        return DFX_OK;
    }
    if (!std::strcmp(name, "nv"))
        return DFX_ERR_NV_DISABLED;
    return DFX_ERR_UNKNOWN_ALGO;
}

const char *dfx_algo_error_message(int status, const char *name, char *buf, size_t buflen) {
    if (!buf || buflen == 0)
        return "";
    if (status == DFX_ERR_NV_DISABLED)
        snprintf(buf, buflen, "NV hardware flow not enabled, pls recompile"); // src/denseflow_gpu.cpp:296
    else if (status == DFX_ERR_UNKNOWN_ALGO)
        snprintf(buf, buflen, "unknown optical algorithm %s", name ? name : ""); // src/denseflow_gpu.cpp:336
    else
        buf[0] = 0;
    return buf;
}

int dfx_create(dfx_handle *out, int device, dfx_algo algo, int width, int height, const dfx_params *params) {
    if (!out)
        return fail(nullptr, DFX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (width < 1 || height < 1 || width > 32768 || height > 32768)
        return fail(nullptr, DFX_ERR_INVALID, "invalid frame size");
    if (algo != DFX_ALGO_TVL1 && algo != DFX_ALGO_FARN && algo != DFX_ALGO_BROX)
        return fail(nullptr, DFX_ERR_INVALID, "invalid algorithm id");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, DFX_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail(nullptr, DFX_ERR_INVALID, "device index out of range");

    dfx_context *c = new dfx_context();
    c->device = device;
    c->algo = algo;
    c->W = width;
    c->H = height;
    if (params)
        c->prm = *params;
    else
        default_params(&c->prm);
    if (const char *g = std::getenv("DFX_GROUP"))
        c->group_override = std::atoi(g);

    int rc = DFX_OK;
    auto init = [&]() -> int {
        HIPCHK(c, hipSetDevice(device));
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_group[0], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_group[1], hipEventDisableTiming));
        HIPCHK(c, hipEventCreate(&c->ev_t0));
        HIPCHK(c, hipEventCreate(&c->ev_t1));
        for (auto &e : c->ev_lvl) {
            HIPCHK(c, hipEventCreate(&e[0]));
            HIPCHK(c, hipEventCreate(&e[1]));
        }
        if (algo == DFX_ALGO_TVL1)
            return create_tvl1(c);
        return fail(c, DFX_ERR_UNSUPPORTED, "algorithm not implemented yet");
    };
    rc = init();
    if (rc != DFX_OK) {
        g_create_error = c->err;
        dfx_destroy(c);
        return rc;
    }
    *out = c;
    return DFX_OK;
}

int dfx_calc(dfx_handle h, const uint8_t *a, size_t a_pitch, const uint8_t *b, size_t b_pitch, float *flow_uv,
             size_t out_pitch) {
    if (!h)
        return DFX_ERR_INVALID;
    if (!a || !b || !flow_uv)
        return fail(h, DFX_ERR_INVALID, "NULL frame or flow pointer");
    if (a_pitch != b_pitch) {
        // stage through the general path with equal pitches: copy rows of b into a temporary
        std::vector<uint8_t> tmp((size_t)h->W * h->H);
        for (int y = 0; y < h->H; ++y)
            std::memcpy(tmp.data() + (size_t)y * h->W, b + (size_t)y * b_pitch, h->W);
        std::vector<uint8_t> tmpa((size_t)h->W * h->H);
        for (int y = 0; y < h->H; ++y)
            std::memcpy(tmpa.data() + (size_t)y * h->W, a + (size_t)y * a_pitch, h->W);
        const uint8_t *fr[2] = {tmpa.data(), tmp.data()};
        float *fl[1] = {flow_uv};
        return dfx_calc_batch(h, fr, h->W, 2, 1, fl, out_pitch);
    }
    const uint8_t *fr[2] = {a, b};
    float *fl[1] = {flow_uv};
    return dfx_calc_batch(h, fr, a_pitch, 2, 1, fl, out_pitch);
}

int dfx_calc_batch(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                   float *const *flows_uv, size_t out_pitch) {
    if (!h)
        return DFX_ERR_INVALID;
    const int M = std::max(n_frames - std::abs(step), 0);
    if (M > 0 && (!frames || !flows_uv))
        return fail(h, DFX_ERR_INVALID, "NULL frames or flows array");
    if (frame_pitch < (size_t)h->W || out_pitch < (size_t)h->W * 8)
        return fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    return calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, flows_uv, out_pitch, nullptr, 0);
}

int dfx_calc_batch_device(dfx_handle h, const uint8_t *d_frames, size_t pitch, size_t frame_stride, int n_frames,
                          int step, float *d_flows, size_t flow_stride_floats) {
    if (!h)
        return DFX_ERR_INVALID;
    const int M = std::max(n_frames - std::abs(step), 0);
    if (M > 0 && (!d_frames || !d_flows))
        return fail(h, DFX_ERR_INVALID, "NULL device frames or flows");
    if (pitch < (size_t)h->W || frame_stride < pitch * (size_t)h->H || flow_stride_floats < (size_t)h->W * h->H * 2)
        return fail(h, DFX_ERR_INVALID, "pitch/stride smaller than a frame");
    return calc_batch_impl(h, nullptr, 0, d_frames, pitch, frame_stride, n_frames, step, nullptr, 0, d_flows,
                           flow_stride_floats);
}

int dfx_get_stats(dfx_handle h, dfx_stats *out) {
    if (!h || !out)
        return DFX_ERR_INVALID;
    *out = h->stats;
    return DFX_OK;
}

void dfx_reset_stats(dfx_handle h) {
    if (h)
        std::memset(&h->stats, 0, sizeof h->stats);
}

const char *dfx_last_error(dfx_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void dfx_destroy(dfx_handle h) {
    if (!h)
        return;
    (void)hipSetDevice(h->device);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    destroy_buffers(h);
    for (auto &e : h->ev_group)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : h->ev_lvl)
        for (auto &x : e)
            if (x)
                (void)hipEventDestroy(x);
    if (h->ev_t0)
        (void)hipEventDestroy(h->ev_t0);
    if (h->ev_t1)
        (void)hipEventDestroy(h->ev_t1);
    if (h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
}

int dfx_device_malloc(dfx_handle h, void **dptr, size_t bytes) {
    if (!h || !dptr)
        return DFX_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMalloc(dptr, bytes));
    return DFX_OK;
}

int dfx_device_free(dfx_handle h, void *dptr) {
    if (!h)
        return DFX_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipFree(dptr));
    return DFX_OK;
}

int dfx_memcpy_h2d(dfx_handle h, void *dst, const void *src, size_t bytes) {
    if (!h)
        return DFX_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return DFX_OK;
}

int dfx_memcpy_d2h(dfx_handle h, void *dst, const void *src, size_t bytes) {
    if (!h)
        return DFX_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return DFX_OK;
}

int dfx_host_alloc(void **ptr, size_t bytes) {
    if (!ptr)
        return DFX_ERR_INVALID;
    return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? DFX_OK : DFX_ERR_HIP;
}

int dfx_host_free(void *ptr) { return hipHostFree(ptr) == hipSuccess ? DFX_OK : DFX_ERR_HIP; }

} // extern "C"
