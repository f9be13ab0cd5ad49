// tests/stub_dfx.cpp — TEST INFRASTRUCTURE, NOT A BACKEND.  A fake of the C ABI of include/dfx.h for the CPU suite: linked
// (instead of libdfx.so) with tools/denseflow.cpp and the host shell into tests/_build/denseflow_stub so that the shell's
// own logic — loader / flow / collector / save threads, FlowBuffer boundaries, joining of short clips, device pipelines,
// naming, .done records, the jpg / png / h5 writers — runs where no GPU exists.  Its "flows" are a made-up function of the
// two frames (below), NOT optical flow; nothing outside tests/ builds or loads this file, and the real CLI
// (build/denseflow) fails with "no HIP device available" on such a machine (tests/test_host_pipeline_stub.py checks both).
// Only what the shell calls is implemented.
//
// STUB_ORACLE=<path to oracle/liboracle.so>: the pair function is the parity ORACLE's (orc_tvl1_calc / orc_farneback_calc /
// orc_brox_calc with the reference's default parameters) instead of the made-up one — BASELINE.json's configs[0] / SURVEY.md
// section 8d "Config 1": the CLI on a CPU, plumbing only (file names, bounding, encoding, "oracle == backend").  Still test
// infrastructure: the oracle is reachable from tests/ only.
#include <unistd.h>
#include <cstdio>
#include <dlfcn.h>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <chrono>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../denseflow_amd/csrc/dfx_plan.h" // the real pair / batch plan: pure host logic
#include "../include/common.h"
#include "../include/dfx.h"
#include "../include/image_io.h"

typedef int (*orc_calc_fn)(const uint8_t *, size_t, const uint8_t *, size_t, int, int, const void *, float *, void *);
typedef int (*orc_calc2_fn)(const uint8_t *, size_t, const uint8_t *, size_t, int, int, const void *, float *);

struct dfx_context {
    int W = 0, H = 0, src_w = 0, src_h = 0;
    int algo = 0;
    void *oracle = nullptr; // dlopen handle of liboracle.so (STUB_ORACLE), or NULL: the made-up pair function
    std::vector<int> next_segments;
    std::string err;
    unsigned long long next_ticket = 1;
};

namespace {
int fail(dfx_context *c, int code, const char *msg) {
    if (c)
        c->err = msg;
    return code;
}

// the fake "flow" of a pair: depends on both frames, on which of them is `a`, and on the position
void fake_flow(const dfx_context *c, const uint8_t *a, const uint8_t *b, float *uv, size_t pitch_floats) {
    if (c->oracle) { // configs[0]: the parity oracle as the "backend" of a CPU plumbing run
        std::vector<float> dense((size_t)c->W * c->H * 2);
        int rc = -1;
        if (c->algo == DFX_ALGO_TVL1) {
            orc_calc_fn f = (orc_calc_fn)dlsym(c->oracle, "orc_tvl1_calc");
            rc = f ? f(a, (size_t)c->W, b, (size_t)c->W, c->W, c->H, nullptr, dense.data(), nullptr) : -1;
        } else {
            orc_calc2_fn f = (orc_calc2_fn)dlsym(c->oracle, c->algo == DFX_ALGO_FARN ? "orc_farneback_calc" : "orc_brox_calc");
            rc = f ? f(a, (size_t)c->W, b, (size_t)c->W, c->W, c->H, nullptr, dense.data()) : -1;
        }
        if (rc != 0)
            std::abort();
        for (int y = 0; y < c->H; ++y)
            std::memcpy(uv + (size_t)y * pitch_floats, dense.data() + (size_t)y * c->W * 2, sizeof(float) * 2 * c->W);
        return;
    }
    for (int y = 0; y < c->H; ++y)
        for (int x = 0; x < c->W; ++x) {
            const int xa = (x + 1) % c->W;
            uv[(size_t)y * pitch_floats + 2 * x] = ((float)b[(size_t)y * c->W + x] - (float)a[(size_t)y * c->W + x]) * 0.125f +
                                                    0.01f * (float)x - 0.3f;
            uv[(size_t)y * pitch_floats + 2 * x + 1] =
                ((float)a[(size_t)y * c->W + xa] - (float)b[(size_t)y * c->W + x]) * 0.0625f + 0.02f * (float)(y % 7);
        }
}

// frames at the engine's size (cv::resize of load_frames_batch when a source format is set), pairs by the real plan
struct Prepared {
    std::vector<std::vector<uint8_t>> frames;
    DfxPairs pairs;
};
int prepare(dfx_context *c, const uint8_t *const *frames, size_t pitch, int n_frames, int step, Prepared &out) {
    if (const char *f = std::getenv("STUB_FAIL_SUBMIT")) // the n-th library call fails (a HIP error in the real library)
        if ((unsigned long long)std::atoll(f) == c->next_ticket) {
            c->next_segments.clear();
            return fail(c, DFX_ERR_HIP, "stub: hipMemcpyAsync failed");
        }
    if (const char *ms = std::getenv("STUB_DELAY_MS")) // a slow "device": lets the loader run ahead (joining tests)
        std::this_thread::sleep_for(std::chrono::milliseconds(std::atoi(ms)));
    std::vector<int> seg;
    seg.swap(c->next_segments);
    if (n_frames < 0 || step == 0)
        return fail(c, DFX_ERR_INVALID, "n_frames must be >= 0 and step non-zero");
    if (seg.empty())
        seg.push_back(n_frames);
    long long total = 0;
    for (int n : seg)
        total += n;
    if (total != n_frames)
        return fail(c, DFX_ERR_INVALID, "dfx_next_segments: the clip lengths do not add up to n_frames");
    out.pairs = dfx_build_pairs(seg, step);
    const int sw = c->src_w ? c->src_w : c->W, sh = c->src_h ? c->src_h : c->H;
    out.frames.resize(n_frames);
    for (int i = 0; i < n_frames; ++i) {
        Mat src(Size(sw, sh), CV_8UC1), dst;
        for (int y = 0; y < sh; ++y)
            std::memcpy(src.ptr<uchar>(y), frames[i] + (size_t)y * pitch, (size_t)sw);
        if (sw != c->W || sh != c->H)
            resizeLinear(src, dst, Size(c->W, c->H));
        else
            dst = src;
        out.frames[i].resize((size_t)c->W * c->H);
        for (int y = 0; y < c->H; ++y)
            std::memcpy(out.frames[i].data() + (size_t)y * c->W, dst.ptr<uchar>(y), (size_t)c->W);
    }
    return DFX_OK;
}
} // namespace

extern "C" {

int dfx_device_count(void) {
    if (std::getenv("STUB_TRACE_DEVICE_COUNT"))
        std::fprintf(stderr, "stub: dfx_device_count in pid %d\n", (int)getpid());
    const char *e = std::getenv("STUB_DEVICES");
    return e ? std::atoi(e) : 1;
}

void dfx_default_params(dfx_params *p) { std::memset(p, 0, sizeof *p); }

int dfx_algo_from_name(const char *name, dfx_algo *out) {
    const std::string n = name ? name : "";
    if (n == "tvl1")
        *out = DFX_ALGO_TVL1;
    else if (n == "farn")
        *out = DFX_ALGO_FARN;
    else if (n == "brox")
        *out = DFX_ALGO_BROX;
    else
        return n == "nv" ? DFX_ERR_NV_DISABLED : DFX_ERR_UNKNOWN_ALGO;
    return DFX_OK;
}

const char *dfx_algo_error_message(int status, const char *name, char *buf, size_t buflen) {
    if (status == DFX_ERR_NV_DISABLED)
        std::snprintf(buf, buflen, "NV hardware flow not enabled, pls recompile");
    else
        std::snprintf(buf, buflen, "unknown optical algorithm %s", name ? name : "");
    return buf;
}

int dfx_create(dfx_handle *out, int device, dfx_algo algo, int width, int height, const dfx_params *) {
    if (device < 0 || device >= dfx_device_count())
        return DFX_ERR_INVALID;
    dfx_context *c = new dfx_context();
    c->W = width, c->H = height, c->algo = (int)algo;
    if (const char *so = std::getenv("STUB_ORACLE")) {
        c->oracle = dlopen(so, RTLD_NOW | RTLD_LOCAL);
        if (!c->oracle) {
            delete c;
            return DFX_ERR_INVALID;
        }
    }
    *out = c;
    return DFX_OK;
}

void dfx_destroy(dfx_handle h) { delete h; }

const char *dfx_last_error(dfx_handle h) { return h ? h->err.c_str() : "stub: dfx_create failed"; }

int dfx_set_source_format(dfx_handle h, int src_width, int src_height, int channels) {
    if (channels != 1)
        return fail(h, DFX_ERR_UNSUPPORTED, "stub: gray sources only");
    h->src_w = src_width, h->src_h = src_height;
    return DFX_OK;
}

int dfx_next_segments(dfx_handle h, const int *seg_frames, int n_segments) {
    h->next_segments.assign(seg_frames, seg_frames + (n_segments > 0 ? n_segments : 0));
    return DFX_OK;
}

int dfx_submit_batch(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step, float *const *flows_uv,
                     size_t out_pitch, uint64_t *ticket) {
    Prepared p;
    const int rc = prepare(h, frames, frame_pitch, n_frames, step, p);
    if (rc != DFX_OK)
        return rc;
    for (int i = 0; i < p.pairs.size(); ++i)
        fake_flow(h, p.frames[dfx_pair_a(p.pairs, i, step)].data(), p.frames[dfx_pair_b(p.pairs, i, step)].data(), flows_uv[i],
                  out_pitch / sizeof(float));
    *ticket = p.pairs.size() ? h->next_ticket++ : 0;
    return DFX_OK;
}

static int bounded_planes(dfx_handle h, const Prepared &p, int step, double lo, double hi, std::vector<Mat> &px, std::vector<Mat> &py) {
    std::vector<float> uv((size_t)h->W * h->H * 2);
    for (int i = 0; i < p.pairs.size(); ++i) {
        fake_flow(h, p.frames[dfx_pair_a(p.pairs, i, step)].data(), p.frames[dfx_pair_b(p.pairs, i, step)].data(), uv.data(),
                  (size_t)h->W * 2);
        Mat flow(Size(h->W, h->H), CV_32FC2), planes[2];
        for (int y = 0; y < h->H; ++y)
            std::memcpy(flow.ptr<float>(y), uv.data() + (size_t)y * h->W * 2, sizeof(float) * 2 * h->W);
        split(flow, planes);
        Mat ix(Size(h->W, h->H), CV_8UC1), iy(Size(h->W, h->H), CV_8UC1);
        convertFlowToImage(planes[0], planes[1], ix, iy, lo, hi);
        px.push_back(ix), py.push_back(iy);
    }
    return DFX_OK;
}

int dfx_submit_batch_u8(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step, double lower_bound,
                        double upper_bound, uint8_t *const *img_x, uint8_t *const *img_y, size_t img_pitch, uint64_t *ticket) {
    Prepared p;
    const int rc = prepare(h, frames, frame_pitch, n_frames, step, p);
    if (rc != DFX_OK)
        return rc;
    std::vector<Mat> px, py;
    bounded_planes(h, p, step, lower_bound, upper_bound, px, py);
    for (size_t i = 0; i < px.size(); ++i)
        for (int y = 0; y < h->H; ++y) {
            std::memcpy(img_x[i] + (size_t)y * img_pitch, px[i].ptr<uchar>(y), (size_t)h->W);
            std::memcpy(img_y[i] + (size_t)y * img_pitch, py[i].ptr<uchar>(y), (size_t)h->W);
        }
    *ticket = px.size() ? h->next_ticket++ : 0;
    return DFX_OK;
}

// the -st=png scheme of the fake: the oracle's restatement when one is loaded (STUB_ORACLE), else the same formulas inline
int dfx_submit_batch_png(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                         uint8_t *const *img_x, uint8_t *const *img_y, size_t img_pitch, double *bounds_xy, uint64_t *ticket) {
    Prepared p;
    const int rc = prepare(h, frames, frame_pitch, n_frames, step, p);
    if (rc != DFX_OK)
        return rc;
    typedef void (*png_fn)(const float *, int, int, uint8_t *, uint8_t *, double *, uint8_t *);
    png_fn orc = h->oracle ? (png_fn)dlsym(h->oracle, "orc_flow_to_png_planes") : nullptr;
    std::vector<float> uv((size_t)h->W * h->H * 2);
    std::vector<uint8_t> x((size_t)h->W * h->H), y((size_t)h->W * h->H);
    for (int i = 0; i < p.pairs.size(); ++i) {
        fake_flow(h, p.frames[dfx_pair_a(p.pairs, i, step)].data(), p.frames[dfx_pair_b(p.pairs, i, step)].data(), uv.data(),
                  (size_t)h->W * 2);
        if (orc) {
            orc(uv.data(), h->W, h->H, x.data(), y.data(), bounds_xy + 2 * i, nullptr);
        } else {
            double mx[2] = {0, 0};
            for (size_t k = 0; k < x.size(); ++k)
                for (int c = 0; c < 2; ++c)
                    mx[c] = std::max(mx[c], (double)std::fabs(uv[2 * k + c]));
            for (int c = 0; c < 2; ++c) {
                double b = std::min(255. * 4, std::ceil((std::min<double>(c ? h->H : h->W, mx[c]) * 128. / 127.) / 4) * 4);
                if ((int)b % 8 == 0)
                    b += 4;
                bounds_xy[2 * i + c] = b;
                const float a = (float)(1. / ((1. / 128.) * b));
                for (size_t k = 0; k < x.size(); ++k) {
                    const float t = uv[2 * k + c] * a;
                    (c ? y : x)[k] = (uint8_t)std::min(255L, std::max(0L, std::lrint((double)(t + 128.f))));
                }
            }
        }
        for (int r = 0; r < h->H; ++r) {
            std::memcpy(img_x[i] + (size_t)r * img_pitch, x.data() + (size_t)r * h->W, (size_t)h->W);
            std::memcpy(img_y[i] + (size_t)r * img_pitch, y.data() + (size_t)r * h->W, (size_t)h->W);
        }
    }
    *ticket = p.pairs.size() ? h->next_ticket++ : 0;
    return DFX_OK;
}

size_t dfx_jpeg_capacity(dfx_handle h) { return h ? (size_t)h->W * h->H + 4096 : 0; }

int dfx_submit_batch_jpeg(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step, double lower_bound,
                          double upper_bound, int quality, uint8_t *const *jpg_x, uint8_t *const *jpg_y, size_t jpg_capacity,
                          uint32_t *size_x, uint32_t *size_y, uint64_t *ticket) {
    if (std::getenv("STUB_JPEG_UNSUPPORTED")) { // the shell's fallback: planes that "do not compress" are encoded on the host
        h->next_segments.clear();
        return fail(h, DFX_ERR_UNSUPPORTED, "stub: JPEG: the batch does not compress below 4 bits per pixel");
    }
    if (std::getenv("STUB_FAST")) { // host-stage timing (scripts/host_stage_rate.py): no work at all here, one constant file
        std::vector<int> seg;
        seg.swap(h->next_segments);
        if (seg.empty())
            seg.push_back(n_frames);
        const int m = dfx_build_pairs(seg, step).size();
        static const vector<uchar> file = [&] {
            Mat flat(Size(h->W, h->H), CV_8UC1);
            std::memset(flat.data(), 128, flat.total());
            vector<uchar> f;
            imencodeJpeg(flat, f, quality);
            return f;
        }();
        for (int i = 0; i < m; ++i) {
            std::memcpy(jpg_x[i], file.data(), file.size());
            std::memcpy(jpg_y[i], file.data(), file.size());
            size_x[i] = size_y[i] = (uint32_t)file.size();
        }
        *ticket = m ? h->next_ticket++ : 0;
        return DFX_OK;
    }
    Prepared p;
    const int rc = prepare(h, frames, frame_pitch, n_frames, step, p);
    if (rc != DFX_OK)
        return rc;
    std::vector<Mat> px, py;
    bounded_planes(h, p, step, lower_bound, upper_bound, px, py);
    for (size_t i = 0; i < px.size(); ++i) {
        vector<uchar> fx, fy;
        imencodeJpeg(px[i], fx, quality); // the device encoder writes these very bytes (tests/test_jpeg_gpu.py)
        imencodeJpeg(py[i], fy, quality);
        if (fx.size() > jpg_capacity || fy.size() > jpg_capacity)
            return fail(h, DFX_ERR_INVALID, "JPEG: jpg_capacity is too small for an encoded plane");
        std::memcpy(jpg_x[i], fx.data(), fx.size());
        std::memcpy(jpg_y[i], fy.data(), fy.size());
        size_x[i] = (uint32_t)fx.size(), size_y[i] = (uint32_t)fy.size();
    }
    *ticket = px.size() ? h->next_ticket++ : 0;
    return DFX_OK;
}

int dfx_wait(dfx_handle h, uint64_t ticket) {
    // STUB_FAIL_WAIT=<ticket>: the tail of that FlowBuffer "fails" (a deferred download error in the real library): the
    // shell must stop without writing its flows, without marking the video done, and without hanging a stage
    if (const char *f = std::getenv("STUB_FAIL_WAIT"))
        if (ticket != 0 && (uint64_t)std::atoll(f) == ticket)
            return fail(h, DFX_ERR_HIP, "stub: deferred download failed");
    return DFX_OK;
}

int dfx_host_alloc(void **ptr, size_t bytes) {
    *ptr = std::malloc(bytes ? bytes : 1);
    return *ptr ? DFX_OK : DFX_ERR_HIP;
}
int dfx_host_free(void *ptr) {
    std::free(ptr);
    return DFX_OK;
}
}
