// brox_engine.cpp — host control of the -a=brox path (replaces cv::cuda::BroxOpticalFlow::calc as called at
// /root/reference/src/denseflow_gpu.cpp:303, :332-334, including the 1/255 convertTo the reference does itself).
// The algorithm is the one defined in oracle/brox_oracle.h (reference parity unpinned).
// Fixed control flow: a pair is a straight sequence of launches, `batch` pairs share each launch
// (grid.z = pair); a frame's pyramid and its five derivative pyramids are built once and serve two pairs.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "brox_kernels.h"
#include "dfx_internal.h"

namespace {

struct BLevel {
    int w, h, pitch;
    long long off;
};

class BroxEngine final : public AlgoEngine {
  public:
    explicit BroxEngine(dfx_context *ctx) : c(ctx) {}
    ~BroxEngine() override { destroy(); }
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
    BroxLevelCtx level_ctx(int l, int nb) const;
    dfx_context *c;
    std::vector<BLevel> lv;
    long long pyr_elems = 0, frame_elems = 0;
    int n_frame_slots = 0;
    float *d_frames = nullptr;
    int *d_frame_slots = nullptr, *h_slots_pinned = nullptr;
    int B = 0;
    int n_cu = 0; // compute units of the device: the streaming SOR runs one persistent workgroup on each
    float *d_planes = nullptr;
    long long plane_stride = 0, slot_stride = 0;
    PairDesc *d_pairs = nullptr, *h_pairs_pinned = nullptr;
    float *d_sor_sink = nullptr; // BroxLevelCtx::sor_sink
    hipEvent_t ev[2] = {nullptr, nullptr};
    uint64_t batch_launches = 0;
};

void BroxEngine::destroy() {
    dfx_free_dev(d_frames);
    dfx_free_dev(d_frame_slots);
    dfx_free_host(h_slots_pinned);
    dfx_free_dev(d_planes);
    dfx_free_dev(d_pairs);
    dfx_free_dev(d_sor_sink);
    dfx_free_host(h_pairs_pinned);
    for (auto &e : ev)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
}

int BroxEngine::create() {
    const dfx_params &p = c->prm;
    if (p.impl < 0 || p.impl > 1)
        return dfx_fail(c, DFX_ERR_INVALID, "brox: impl must be 0 (tuned) or 1 (simple)");
    if (!(p.brox_scale_factor > 0.f && p.brox_scale_factor < 1.f) || !(p.brox_alpha > 0.f) ||
        p.brox_inner_iterations < 0 || p.brox_outer_iterations < 1 || p.brox_solver_iterations < 0)
        return dfx_fail(c, DFX_ERR_INVALID, "invalid Brox parameters");
    {
        int dev = 0;
        HIPCHK(c, hipGetDevice(&dev));
        HIPCHK(c, hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    // pyramid sizes: scale accumulated in float, ceilf, until a side is <= 15 px or outer_iterations levels
    {
        float scale = 1.0f;
        int pw = c->W, ph = c->H;
        long long off = 0;
        lv.push_back(BLevel{c->W, c->H, dfx_round_up(c->W, 64), 0});
        off += (long long)lv[0].pitch * c->H;
        while (pw > 15 && ph > 15 && (int)lv.size() < p.brox_outer_iterations && lv.size() < 128) {
            scale *= p.brox_scale_factor;
            const int w = (int)std::ceil((float)c->W * scale), h = (int)std::ceil((float)c->H * scale);
            lv.push_back(BLevel{w, h, dfx_round_up(w, 64), off});
            off += (long long)lv.back().pitch * h;
            pw = w;
            ph = h;
        }
        pyr_elems = off;
        frame_elems = off * BROX_FP_COUNT;
    }
    plane_stride = (long long)lv[0].pitch * c->H;
    slot_stride = plane_stride * BROX_PL_COUNT;
    B = p.max_batch;
    if (B <= 0) {
        const long long px0 = (long long)c->W * c->H;
        B = (int)std::max<long long>(1, std::min<long long>(DFX_MAX_BATCH, (256LL << 20) / std::max<long long>(px0, 1)));
    }
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    const size_t per_pair = (size_t)slot_stride * 4 + (size_t)frame_elems * 4 + (size_t)c->W * c->H * 9;
    while (B > 1 && per_pair * (size_t)(B + 2) > free_b / 2)
        B /= 2;
    HIPCHK(c, hipMalloc(&d_planes, (size_t)slot_stride * B * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_pairs, sizeof(PairDesc) * B));
    HIPCHK(c, hipMalloc(&d_sor_sink, sizeof(float) * 16 * (size_t)std::max(n_cu, 1)));
    HIPCHK(c, hipHostMalloc(&h_pairs_pinned, sizeof(PairDesc) * B, hipHostMallocDefault));
    HIPCHK(c, hipEventCreateWithFlags(&ev[0], dfx_event_flags(c, true)));
    HIPCHK(c, hipEventCreateWithFlags(&ev[1], dfx_event_flags(c, true)));
    return ensure_frame_slots(B + 1);
}

int BroxEngine::ensure_frame_slots(int need) {
    if (need <= n_frame_slots)
        return DFX_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfx_free_dev(d_frames);
    dfx_free_dev(d_frame_slots);
    dfx_free_host(h_slots_pinned);
    HIPCHK(c, hipMalloc(&d_frames, (size_t)need * frame_elems * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_frame_slots, sizeof(int) * need));
    HIPCHK(c, hipHostMalloc(&h_slots_pinned, sizeof(int) * need, hipHostMallocDefault));
    n_frame_slots = need;
    return DFX_OK;
}

int BroxEngine::build_frames(const unsigned char *d_src, long long src_frame_stride, long long src_pitch, int n,
                             const int *h_slots) {
    if (n <= 0)
        return DFX_OK;
    std::memcpy(h_slots_pinned, h_slots, sizeof(int) * n);
    HIPCHK(c, hipMemcpyAsync(d_frame_slots, h_slots_pinned, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    const float a255 = (float)(1.0 / 255.0); // the reference's convertTo(CV_32F, 1.0/255.0) (:332-333)
    brox_launch_u8_to_f32(c->stream, d_src, src_frame_stride, src_pitch, d_frame_slots, n, d_frames, frame_elems,
                          lv[0].w, lv[0].h, lv[0].pitch, a255);
    for (size_t l = 1; l < lv.size(); ++l)
        brox_launch_downsample(c->stream, d_frames, frame_elems, d_frame_slots, n, lv[l - 1].off, lv[l - 1].w,
                               lv[l - 1].h, lv[l - 1].pitch, lv[l].off, lv[l].w, lv[l].h, lv[l].pitch,
                               c->prm.brox_scale_factor);
    for (size_t l = 0; l < lv.size(); ++l) {
        const BLevel &L = lv[l];
        auto P = [&](int plane) { return (long long)plane * pyr_elems + L.off; };
        brox_launch_deriv(c->stream, d_frames, frame_elems, d_frame_slots, n, P(BROX_FP_I), P(BROX_FP_DX), L.w, L.h, L.pitch, 0);
        brox_launch_deriv(c->stream, d_frames, frame_elems, d_frame_slots, n, P(BROX_FP_I), P(BROX_FP_DY), L.w, L.h, L.pitch, 1);
        brox_launch_deriv(c->stream, d_frames, frame_elems, d_frame_slots, n, P(BROX_FP_DX), P(BROX_FP_DXX), L.w, L.h, L.pitch, 0);
        brox_launch_deriv(c->stream, d_frames, frame_elems, d_frame_slots, n, P(BROX_FP_DY), P(BROX_FP_DYY), L.w, L.h, L.pitch, 1);
        brox_launch_deriv(c->stream, d_frames, frame_elems, d_frame_slots, n, P(BROX_FP_DX), P(BROX_FP_DXY), L.w, L.h, L.pitch, 1);
    }
    c->stats.kernel_launches += 1 + (lv.size() - 1) + 5 * lv.size();
    return DFX_OK;
}

BroxLevelCtx BroxEngine::level_ctx(int l, int nb) const {
    BroxLevelCtx x;
    std::memset(&x, 0, sizeof x);
    x.w = lv[l].w;
    x.h = lv[l].h;
    x.pitch = lv[l].pitch;
    x.lvl_off = lv[l].off;
    x.frames = d_frames;
    x.frame_stride = frame_elems;
    x.pyr_elems = pyr_elems;
    x.planes = d_planes;
    x.plane_stride = plane_stride;
    x.slot_stride = slot_stride;
    x.pairs = d_pairs;
    x.n_pairs = nb;
    x.alpha = c->prm.brox_alpha;
    x.gamma = c->prm.brox_gamma;
    x.omega = 1.99f;
    x.sor_sink = d_sor_sink;
    x.sor_progress = (c->prm.variant & DFX_VAR_BROX_SOR_PROGRESS) ? 1 : 0;
    x.sor_stream = (c->prm.variant & (DFX_VAR_BROX_SOR_PROGRESS | DFX_VAR_BROX_SOR_PER_TILE)) ? 0 : n_cu;
    return x;
}

int BroxEngine::run_pairs(int nb, const PairDesc *h_pairs, float *d_out, long long out_stride) {
    std::memcpy(h_pairs_pinned, h_pairs, sizeof(PairDesc) * nb);
    HIPCHK(c, hipMemcpyAsync(d_pairs, h_pairs_pinned, sizeof(PairDesc) * nb, hipMemcpyHostToDevice, c->stream));
    const dfx_params &p = c->prm;
    int uv = 0;
    batch_launches = 0;
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    for (int l = (int)lv.size() - 1; l >= 0; --l) {
        const BroxLevelCtx x = level_ctx(l, nb);
        brox_launch_level_init(c->stream, x, uv, l == (int)lv.size() - 1);
        int ds = 0; // du/dv set holding the current increment
        for (int in = 0; in < p.brox_inner_iterations; ++in) {
            brox_launch_stage1(c->stream, x, uv, ds);
            if (p.impl == 1) { // simple form: stage 2 as its own launch, then one launch per half sweep, in place
                brox_launch_stage2(c->stream, x);
                for (int si = 0; si < p.brox_solver_iterations; ++si) {
                    brox_launch_sor(c->stream, x, uv, ds, 0);
                    brox_launch_sor(c->stream, x, uv, ds, 1);
                }
            } else {
                const int S = brox_fused_sweeps();
                for (int si = 0; si < p.brox_solver_iterations; si += S) {
                    brox_launch_sor_fused(c->stream, x, uv, ds, std::min(S, p.brox_solver_iterations - si));
                    ds ^= 1; // it wrote the other set
                }
            }
        }
        brox_launch_add_increment(c->stream, x, uv, ds);
        batch_launches += 2 + (uint64_t)p.brox_inner_iterations *
                                  (1 + (p.impl == 1 ? 1 + 2 * p.brox_solver_iterations
                                                    : (p.brox_solver_iterations + brox_fused_sweeps() - 1) / brox_fused_sweeps()));
        if (l > 0) {
            brox_launch_prolongate(c->stream, x, uv, lv[l - 1].w, lv[l - 1].h, lv[l - 1].pitch, p.brox_scale_factor,
                                   1.0f / p.brox_scale_factor);
            uv ^= 1;
            batch_launches += 1;
        } else {
            brox_launch_merge(c->stream, x, uv, d_out, out_stride);
            batch_launches += 1;
        }
    }
    HIPCHK(c, hipEventRecord(ev[1], c->stream));
    c->stats.kernel_launches += batch_launches;
    return DFX_OK;
}

// Byte model (defined in DESIGN.md §4): per level N px, per inner iteration stage 1+2 move 100 B/px and every
// red+black SOR sweep 52 B/px (13 float planes touched once).
int BroxEngine::account(int nb) {
    dfx_stats &st = c->stats;
    const dfx_params &p = c->prm;
    double bytes = 0, sor_bytes = 0;
    for (const BLevel &L : lv) {
        const double N = (double)L.w * L.h;
        const double s = N * p.brox_inner_iterations * p.brox_solver_iterations * 52.0;
        bytes += N * p.brox_inner_iterations * 100.0 + s + N * 40.0;
        sor_bytes += s;
    }
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, ev[0], ev[1]));
    st.step_ms += ms;
    st.step_launches += batch_launches;
    st.algorithmic_bytes += bytes * nb;
    st.step_algorithmic_bytes += bytes * nb;
    (void)sor_bytes;
    st.pairs += (uint64_t)nb;
    st.levels = (int)std::min<size_t>(lv.size(), DFX_MAX_LEVELS);
    for (int l = 0; l < st.levels; ++l) {
        st.level_w[l] = lv[l].w;
        st.level_h[l] = lv[l].h;
    }
    return DFX_OK;
}

} // namespace

AlgoEngine *dfx_make_brox_engine(dfx_context *c) { return new BroxEngine(c); }
