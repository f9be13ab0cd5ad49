// farneback_engine.cpp — host control of the -a=farn path (replaces cv::cuda::FarnebackOpticalFlow::calc
// as called at /root/reference/src/denseflow_gpu.cpp:329; algorithm: SURVEY.md Appendix B).
//
// Control flow is fixed (no data-dependent exit), so a pair is a straight sequence of launches with
// no host synchronisation; `batch` pairs share every launch (grid.z = pair).  Per-frame work (u8 ->
// f32, blur+resize per level, polynomial expansion) is done once per frame and reused by both pairs
// the frame belongs to.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstring>

#include "dfx_internal.h"
#include "farneback_kernels.h"

namespace {

constexpr int kMinSize = 32; // upstream MIN_SIZE

struct FLevel {
    FarnLevelGeom g;
    double sigma;
    int half;    // Gaussian pre-blur half width (smoothSize / 2)
    int ker_off; // offset of this level's taps (centre first) in d_gker
    float ifx, ify;
};

// B.3: polynomial-expansion constants (identical to upstream FarnebackPrepareGaussian)
void prepare_poly(int n, double sigma, FarnPolyConsts *out) {
    std::vector<float> buf(n * 6 + 3);
    float *g = buf.data() + n, *xg = g + n * 2 + 1, *xxg = xg + n * 2 + 1;
    if (sigma < FLT_EPSILON)
        sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    // normal matrix of the basis {1, x, y, x^2, y^2, xy} under the applicability g; only four of its
    // entries are distinct, so the 6x6 inverse reduces to a 3x3 block {1, x^2, y^2} plus diagonals
    double a = 0, bq = 0, cq = 0, d = 0;
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            a += g[y] * g[x];
            bq += g[y] * g[x] * x * x;
            cq += g[y] * g[x] * x * x * x * x;
            d += g[y] * g[x] * x * x * y * y;
        }
    // block [[a, b, b], [b, c, d], [b, d, c]] over (1, x^2, y^2): closed-form inverse entries
    const double det = a * (cq * cq - d * d) - 2.0 * bq * bq * (cq - d);
    const double inv03 = -bq * (cq - d) / det;   // (1, x^2)
    const double inv33 = (a * cq - bq * bq) / det; // (x^2, x^2)
    out->ig11 = (float)(1.0 / bq);
    out->ig03 = (float)inv03;
    out->ig33 = (float)inv33;
    out->ig55 = (float)(1.0 / d);
    for (int i = 0; i <= n; ++i) {
        out->g[i] = g[i];
        out->xg[i] = xg[i];
        out->xxg[i] = xxg[i];
    }
}

// B.6: cv::getGaussianKernel(ksize, sigma, CV_32F); returns taps centre-first (k[0] = centre)
bool gaussian_taps(int n, double sigma, std::vector<float> &half_out) {
    if (n < 1 || !(n & 1))
        return false;
    std::vector<double> k(n);
    bool fixed = false;
    if (sigma <= 0) {
        static const double t3[] = {0.25, 0.5, 0.25};
        static const double t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
        static const double t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
        const double *t = n == 3 ? t3 : n == 5 ? t5 : n == 7 ? t7 : nullptr;
        if (n == 1) {
            k[0] = 1.0;
            fixed = true;
        } else if (t) {
            std::copy(t, t + n, k.begin());
            fixed = true;
        }
    }
    if (!fixed) {
        const double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
        const double scale2x = -0.125 / (sx * sx);
        const int n2 = (n - 1) / 2;
        double sum = 0;
        for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
            k[i] = std::exp((double)(x * x) * scale2x);
            sum += k[i];
        }
        sum = sum * 2 + 1.0;
        const double mul = 1.0 / sum;
        for (int i = 0; i < n2; ++i)
            k[n - 1 - i] = k[i] = (double)(float)(k[i] * mul);
        k[n2] = (double)(float)mul;
    }
    const int half = n / 2;
    half_out.resize(half + 1);
    for (int j = 0; j <= half; ++j)
        half_out[j] = (float)k[half + j];
    return true;
}

class FarnebackEngine final : public AlgoEngine {
  public:
    explicit FarnebackEngine(dfx_context *ctx) : c(ctx) {}
    ~FarnebackEngine() override { destroy(); }
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
    dfx_context *c;
    int nlev = 0; // levels 0..nlev-1 (nlev = numLevelsCropped + 1)
    FLevel lv[DFX_LVL_MAX];
    long long frame_elems = 0;
    int pitch0 = 0;
    FarnPolyConsts pc{};
    float *d_gker = nullptr;

    int n_frame_slots = 0;
    float *d_R = nullptr;
    int *d_frame_slots = nullptr;
    int *h_slots_pinned = nullptr;
    // scratch for frame preparation, sized for n_frame_slots frames
    float *d_f32 = nullptr, *d_tmpv = nullptr, *d_pyr = nullptr;
    int skip_zero_weights = 1, polyexp_rows = 16; // frame-preparation forms, fixed when the engine is created
    bool m_on_chip = true;                        // the default iteration kernel (M recomputed, never in HBM)

    int B = 0;
    float *d_planes = nullptr;
    long long plane_stride = 0, slot_stride = 0;
    PairDesc *d_pairs = nullptr, *h_pairs_pinned = nullptr;
    hipEvent_t ev_it[DFX_LVL_MAX][2] = {};
};

void FarnebackEngine::destroy() {
    dfx_free_dev(d_gker);
    dfx_free_dev(d_R);
    dfx_free_dev(d_frame_slots);
    dfx_free_host(h_slots_pinned);
    dfx_free_dev(d_f32);
    dfx_free_dev(d_tmpv);
    dfx_free_dev(d_pyr);
    dfx_free_dev(d_planes);
    dfx_free_dev(d_pairs);
    dfx_free_host(h_pairs_pinned);
    for (auto &e : ev_it)
        for (auto &x : e)
            if (x) {
                (void)hipEventDestroy(x);
                x = nullptr;
            }
}

int FarnebackEngine::create() {
    const dfx_params &p = c->prm;
    if (p.impl < 0 || p.impl > 1)
        return dfx_fail(c, DFX_ERR_INVALID, "farn: impl must be 0 (tuned) or 1 (simple)");
    // cross-check forms of the frame preparation (dfx_params.variant; same bits both ways)
    skip_zero_weights = (p.variant & DFX_VAR_FARN_EVAL_ZERO_TAPS) ? 0 : farn_skip_zero_weights_default();
    polyexp_rows = (p.variant & DFX_VAR_FARN_POLY_ONE_ROW) ? 0 : farn_polyexp_rows_default();
    if (p.farn_poly_n != 5)
        return dfx_fail(c, DFX_ERR_UNSUPPORTED, "Farneback: only polyN = 5 (the reference's default) is built");
    if (p.farn_flags != 0)
        return dfx_fail(c, DFX_ERR_UNSUPPORTED, "Farneback: only flags = 0 (box-filter update, the reference's default)");
    if (p.farn_win_size < 1 || !(p.farn_win_size & 1) || p.farn_win_size / 2 > 8)
        return dfx_fail(c, DFX_ERR_UNSUPPORTED, "Farneback: winSize must be odd and <= 17");
    if (p.farn_num_levels < 0 || p.farn_num_levels >= DFX_LVL_MAX || p.farn_num_iters < 1 ||
        !(p.farn_pyr_scale > 0.0 && p.farn_pyr_scale < 1.0))
        return dfx_fail(c, DFX_ERR_INVALID, "invalid Farneback parameters");

    const int W = c->W, H = c->H;
    pitch0 = dfx_round_up(W, 64);
    // B.2: crop levels whose size would drop below MIN_SIZE
    double scale = 1;
    int cropped = 0;
    for (; cropped < p.farn_num_levels; cropped++) {
        scale *= p.farn_pyr_scale;
        if (W * scale < kMinSize || H * scale < kMinSize)
            break;
    }
    nlev = cropped + 1;
    std::vector<float> all_taps;
    long long off = 0;
    for (int k = 0; k < nlev; ++k) {
        scale = 1;
        for (int i = 0; i < k; i++)
            scale *= p.farn_pyr_scale;
        FLevel &L = lv[k];
        L.sigma = (1. / scale - 1) * 0.5;
        int smooth = dfx_cv_round(L.sigma * 5) | 1;
        smooth = std::max(smooth, 3);
        L.half = smooth / 2;
        L.g.w = dfx_cv_round(W * scale);
        L.g.h = dfx_cv_round(H * scale);
        L.g.pitch = dfx_round_up(L.g.w, 64);
        L.g.r_off = off;
        off += 5LL * L.g.pitch * L.g.h;
        L.ifx = (float)(1.0 / ((double)L.g.w / (double)W)); // dsize given (E.1)
        L.ify = (float)(1.0 / ((double)L.g.h / (double)H));
        std::vector<float> taps;
        if (!gaussian_taps(smooth, L.sigma, taps))
            return dfx_fail(c, DFX_ERR_INVALID, "Farneback: bad Gaussian kernel size");
        L.ker_off = (int)all_taps.size();
        all_taps.insert(all_taps.end(), taps.begin(), taps.end());
    }
    frame_elems = off;
    prepare_poly(p.farn_poly_n, p.farn_poly_sigma, &pc);
    HIPCHK(c, hipMalloc(&d_gker, sizeof(float) * all_taps.size()));
    HIPCHK(c, hipMemcpy(d_gker, all_taps.data(), sizeof(float) * all_taps.size(), hipMemcpyHostToDevice));

    plane_stride = (long long)pitch0 * H;
    // The default iteration kernel keeps M on chip: a pair slot is its two flow sets (4 planes, 33 MB at 1080p).  The
    // M-in-HBM kernels (another window size, impl = 1, DFX_VAR_FARN_M_IN_HBM) need the two M sets as well (14 planes).
    m_on_chip = p.impl == 0 && p.farn_win_size / 2 == 6 && !(p.variant & DFX_VAR_FARN_M_IN_HBM);
    slot_stride = plane_stride * (m_on_chip ? (int)FARN_PL_M0 : (int)FARN_PL_COUNT);
    B = p.max_batch;
    if (B <= 0) {
        const long long px0 = (long long)W * H;
        B = (int)std::max<long long>(1, std::min<long long>(DFX_MAX_BATCH, (256LL << 20) / std::max<long long>(px0, 1)));
    }
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    const size_t per_pair = (size_t)slot_stride * 4 + (size_t)frame_elems * 4 + (size_t)plane_stride * 16 + (size_t)W * H * 9;
    while (B > 1 && per_pair * (size_t)(B + 2) > free_b / 2)
        B /= 2;
    HIPCHK(c, hipMalloc(&d_planes, (size_t)slot_stride * B * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_pairs, sizeof(PairDesc) * B));
    HIPCHK(c, hipHostMalloc(&h_pairs_pinned, sizeof(PairDesc) * B, hipHostMallocDefault));
    for (auto &e : ev_it) {
        HIPCHK(c, hipEventCreateWithFlags(&e[0], dfx_event_flags(c, true)));
        HIPCHK(c, hipEventCreateWithFlags(&e[1], dfx_event_flags(c, true)));
    }
    return ensure_frame_slots(B + 1);
}

int FarnebackEngine::ensure_frame_slots(int need) {
    if (need <= n_frame_slots)
        return DFX_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfx_free_dev(d_R);
    dfx_free_dev(d_frame_slots);
    dfx_free_host(h_slots_pinned);
    dfx_free_dev(d_f32);
    dfx_free_dev(d_tmpv);
    dfx_free_dev(d_pyr);
    HIPCHK(c, hipMalloc(&d_R, (size_t)need * frame_elems * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_frame_slots, sizeof(int) * need));
    HIPCHK(c, hipHostMalloc(&h_slots_pinned, sizeof(int) * need, hipHostMallocDefault));
    if (!skip_zero_weights) // only the cross-check chain converts the frames to a float plane first
        HIPCHK(c, hipMalloc(&d_f32, (size_t)need * plane_stride * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_tmpv, (size_t)need * plane_stride * 2 * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_pyr, (size_t)need * plane_stride * sizeof(float)));
    n_frame_slots = need;
    return DFX_OK;
}

int FarnebackEngine::build_frames(const unsigned char *d_src, long long src_frame_stride, long long src_pitch, int n,
                                  const int *h_slots) {
    if (n <= 0)
        return DFX_OK;
    std::memcpy(h_slots_pinned, h_slots, sizeof(int) * n);
    HIPCHK(c, hipMemcpyAsync(d_frame_slots, h_slots_pinned, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    const int W = c->W, H = c->H;
    // The vertical blur reads the 8-bit frames themselves unless the cross-check form is asked for
    // (DFX_VAR_FARN_EVAL_ZERO_TAPS: the round-1 chain with its separate convertTo pass; same bits either way).
    const bool from_u8 = skip_zero_weights != 0;
    if (!from_u8)
        farn_launch_u8_to_f32(c->stream, d_src, src_frame_stride, src_pitch, n, d_f32, plane_stride, W, H, pitch0);
    for (int k = nlev - 1; k >= 0; --k) {
        const FLevel &L = lv[k];
        if (from_u8)
            farn_launch_blur_v_u8(c->stream, d_src, src_frame_stride, src_pitch, n, W, H, pitch0, L.g.h, L.ify,
                                  d_gker + L.ker_off, L.half, d_tmpv, plane_stride * 2, skip_zero_weights);
        else
            farn_launch_blur_v(c->stream, d_f32, plane_stride, n, W, H, pitch0, L.g.h, L.ify, d_gker + L.ker_off, L.half,
                               d_tmpv, plane_stride * 2, skip_zero_weights);
        farn_launch_blur_h_resize(c->stream, d_tmpv, plane_stride * 2, n, W, H, pitch0, L.g.w, L.g.h, L.g.pitch, L.ifx,
                                  L.ify, d_gker + L.ker_off, L.half, d_pyr, plane_stride, skip_zero_weights);
        farn_launch_polyexp(c->stream, d_pyr, plane_stride, n, d_frame_slots, d_R, frame_elems, L.g, pc, polyexp_rows);
    }
    c->stats.kernel_launches += (from_u8 ? 0 : 1) + 3 * nlev;
    return DFX_OK;
}

int FarnebackEngine::run_pairs(int nb, const PairDesc *h_pairs, float *d_out, long long out_stride) {
    std::memcpy(h_pairs_pinned, h_pairs, sizeof(PairDesc) * nb);
    HIPCHK(c, hipMemcpyAsync(d_pairs, h_pairs_pinned, sizeof(PairDesc) * nb, hipMemcpyHostToDevice, c->stream));
    const dfx_params &p = c->prm;
    const int half = p.farn_win_size / 2;
    const float box_inv = 1.f / (float)((1 + 2 * half) * (1 + 2 * half));
    const float up = (float)(1. / p.farn_pyr_scale);
    FarnPairCtx x;
    std::memset(&x, 0, sizeof x);
    x.frame_R = d_R;
    x.frame_stride = frame_elems;
    x.planes = d_planes;
    x.plane_stride = plane_stride;
    x.slot_stride = slot_stride;
    x.pairs = d_pairs;
    x.n_pairs = nb;
    // M recomputed inside the iteration kernel (round 4) unless the window is not the reference's 13 or a cross-check
    // form is asked for
    const bool fused = m_on_chip;
    if (fused) {
        // One launch per iteration and nothing else: the first iteration of a level up-samples the coarser level's flow
        // itself (zero at the coarsest), the last one of level 0 writes the caller's interleaved rows.  The flow
        // ping-pongs between its two plane sets; `cur` = the set the latest flow is in.
        int cur = 0;
        for (int k = nlev - 1; k >= 0; --k) {
            x.L = lv[k].g;
            const bool top = k == nlev - 1;
            const FarnLevelGeom P = top ? lv[k].g : lv[k + 1].g;
            const float ifx = top ? 0.f : (float)(1.0 / ((double)x.L.w / (double)P.w));
            const float ify = top ? 0.f : (float)(1.0 / ((double)x.L.h / (double)P.h));
            HIPCHK(c, hipEventRecord(ev_it[k][0], c->stream));
            for (int it = 0; it < p.farn_num_iters; ++it) {
                const bool last = k == 0 && it == p.farn_num_iters - 1;
                float *merged = last ? d_out : nullptr;
                if (it == 0)
                    farn_launch_iter_stream_init(c->stream, x, cur, cur ^ 1, box_inv, merged, out_stride, P.w, P.h, P.pitch, ifx,
                                                 ify, up, top ? 1 : 0);
                else
                    farn_launch_iter_stream(c->stream, x, cur, cur ^ 1, box_inv, merged, out_stride);
                cur ^= 1;
            }
            HIPCHK(c, hipEventRecord(ev_it[k][1], c->stream));
            c->stats.kernel_launches += p.farn_num_iters;
        }
        return DFX_OK;
    }
    int set = (nlev - 1) & 1; // flow set the level's iterations run in (the previous level ended in the other one)
    for (int k = nlev - 1; k >= 0; --k) {
        x.L = lv[k].g;
        if (k == nlev - 1) {
            farn_launch_init_flow(c->stream, x, set, 0, 0, 0, 0.f, 0.f, 0.f, 1);
        } else {
            const FarnLevelGeom &P = lv[k + 1].g;
            const float ifx = (float)(1.0 / ((double)x.L.w / (double)P.w));
            const float ify = (float)(1.0 / ((double)x.L.h / (double)P.h));
            farn_launch_init_flow(c->stream, x, set, P.w, P.h, P.pitch, ifx, ify, up, 0); // reads set ^ 1
        }
        farn_launch_update_matrices(c->stream, x, set, 0);
        int m_src = 0;
        HIPCHK(c, hipEventRecord(ev_it[k][0], c->stream));
        for (int it = 0; it < p.farn_num_iters; ++it) {
            const int dm = it < p.farn_num_iters - 1;
            farn_launch_iteration(c->stream, x, set, m_src, half, box_inv, dm, c->prm.impl);
            if (dm)
                m_src ^= 1;
        }
        HIPCHK(c, hipEventRecord(ev_it[k][1], c->stream));
        c->stats.kernel_launches += 2 + p.farn_num_iters;
        set ^= 1; // the next (finer) level is initialised into the other set, from the one this level ended in
    }
    set ^= 1; // the set level 0 ended in
    farn_launch_merge(c->stream, x, set, d_out, out_stride);
    c->stats.kernel_launches += 1;
    return DFX_OK;
}

// SURVEY.md §8d Farneback byte model, per pair as the reference executes it
int FarnebackEngine::account(int nb) {
    dfx_stats &st = c->stats;
    const dfx_params &p = c->prm;
    const double N0 = (double)c->W * c->H;
    double bytes = 34.0 * N0, it_bytes = 0;
    for (int k = 0; k < nlev; ++k) {
        const double Nk = (double)lv[k].g.w * lv[k].g.h;
        bytes += 2.0 * (8.0 * N0 + 4.0 * std::min(N0, 4.0 * Nk) + 4.0 * Nk + 24.0 * Nk);
        bytes += 68.0 * Nk + 16.0 * Nk;
        const double itb = p.farn_num_iters * (40.0 + 28.0) * Nk + (p.farn_num_iters - 1) * 68.0 * Nk;
        bytes += itb;
        it_bytes += itb;
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, ev_it[k][0], ev_it[k][1]));
        st.step_ms += ms;
        st.level_ms[k] += ms;
        st.step_launches += (uint64_t)p.farn_num_iters;
        st.level_launches[k] += (uint64_t)p.farn_num_iters;
    }
    st.algorithmic_bytes += bytes * nb;
    st.step_algorithmic_bytes += it_bytes * nb;
    st.pairs += (uint64_t)nb;
    st.levels = nlev;
    for (int k = 0; k < nlev; ++k) {
        st.level_w[k] = lv[k].g.w;
        st.level_h[k] = lv[k].g.h;
    }
    return DFX_OK;
}

} // namespace

AlgoEngine *dfx_make_farneback_engine(dfx_context *c) { return new FarnebackEngine(c); }
