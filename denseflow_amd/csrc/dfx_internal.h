// dfx_internal.h — host-side plumbing shared by the C ABI (dfx_api.cpp) and the per-algorithm
// engines (tvl1_engine.cpp, farneback_engine.cpp).  Nothing here crosses the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dfx.h"
#include "dfx_device.h"
#include "dfx_helper.h"

// An algorithm engine owns every device buffer of one handle.  The common driver (dfx_api.cpp)
// walks a FlowBuffer in batches of `batch()` pairs, keeps the per-frame derived data (pyramids,
// polynomial expansions) in a ring of frame slots so each frame is prepared once, and calls:
//     build_frames  - prepare `n` new frames (u8 pixels already on the device) into the given slots
//     run_pairs     - compute `nb` flows (pair -> frame-slot descriptors) into d_out
//     account       - fold the finished batch into dfx_stats (after the stream is synchronised)
struct dfx_context;

// Automatic batch (dfx_params.max_batch = 0): as many pairs as 256 Mpx of level-0 pixels hold (129 at 1080p), at most
// this many.  Small frames reach it: 2048 pairs of 224 x 224 are 103 Mpx — 0.4 of the 1080p batch — and a FlowBuffer
// only fills such a batch when it joins several clips (dfx_next_segments).
constexpr long long DFX_MAX_BATCH = 2048;

class AlgoEngine {
  public:
    virtual ~AlgoEngine() {}
    virtual int create() = 0;
    virtual int batch() const = 0;
    virtual int ensure_frame_slots(int need) = 0;
    virtual int frame_slots() const = 0;
    virtual int build_frames(const unsigned char *d_src, long long src_frame_stride, long long src_pitch, int n,
                             const int *h_slots) = 0;
    virtual int run_pairs(int nb, const PairDesc *h_pairs, float *d_out, long long out_stride) = 0;
    virtual int account(int nb) = 0; // reads per-batch event timers; stream is idle
};

struct dfx_context {
    int device = 0;
    dfx_algo algo = DFX_ALGO_TVL1;
    int W = 0, H = 0;
    dfx_params prm{};
    // Last error text.  dfx_wait may run on a collector thread beside the owner's dfx_submit_*, and both may fail:
    // every access goes through set_err / get_err.
    std::string err_;
    mutable std::mutex err_mtx;
    void set_err(const std::string &m) {
        std::lock_guard<std::mutex> lock(err_mtx);
        err_ = m;
    }
    std::string get_err() const {
        std::lock_guard<std::mutex> lock(err_mtx);
        return err_;
    }
    // format of the frames handed to the calc entry points (dfx_set_source_format); 0 = the handle's own W x H gray
    int src_w = 0, src_h = 0, src_ch = 1;
    bool prepares() const { return src_w > 0; }
    int in_w() const { return prepares() ? src_w : W; }   // width / height / bytes per row of an input frame
    int in_h() const { return prepares() ? src_h : H; }
    size_t in_row_bytes() const { return (size_t)in_w() * (prepares() ? src_ch : 1); }

    hipStream_t stream = nullptr;      // compute
    hipStream_t copy_stream = nullptr; // host -> device copies of the host-pointer entry points
    hipStream_t d2h_stream = nullptr;  // device -> host copies: uploads of the next batch / FlowBuffer overlap them
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    hipEvent_t ev_block = nullptr; // dfx_params.blocking_sync: the event dfx_stream_wait sleeps on
    hipEvent_t ev_h2d[2] = {nullptr, nullptr};     // frames of batch i are in staging set i&1
    hipEvent_t ev_compute[2] = {nullptr, nullptr}; // flows of batch i are in staging set i&1
    hipEvent_t ev_d2h[2] = {nullptr, nullptr};     // flow staging set i&1 has been copied out

    AlgoEngine *engine = nullptr;

    // staging shared by every engine
    // host-mode staging, two sets each so that copies of batch i+-1 overlap the compute of batch i
    unsigned char *d_u8[2] = {nullptr, nullptr}; // u8_slots dense W*H frames per set
    int u8_slots = 0;
    float *d_flow_out[2] = {nullptr, nullptr};   // flow_slots dense H*W*2 flows per set
    int flow_slots = 0;
    unsigned char *d_src[2] = {nullptr, nullptr}; // source-format frames before preparation (src_slots per set)
    int src_slots = 0;
    size_t src_frame_bytes = 0;
    // page-locked bounce buffers for FlowBuffers of small frames: one copy per batch instead of one per frame
    unsigned char *h_in[2] = {nullptr, nullptr}, *h_out[2] = {nullptr, nullptr};
    size_t h_in_bytes = 0, h_out_bytes = 0;
    unsigned char *d_img[2] = {nullptr, nullptr}; // bounded output: img_slots x planes, then img_slots y planes
    int img_slots = 0;
    // the -st=png scheme (quantize_kernels.hip): extrema / scale scratch, and per staging parity the adaptive bounds of a
    // batch in mapped page-locked memory (the kernel writes them, the host reads them after the batch's stream wait)
    void *d_png_scratch = nullptr;
    double *h_png_bounds[2] = {nullptr, nullptr}, *d_png_bounds[2] = {nullptr, nullptr};
    int png_slots = 0;
    // device JPEG encoder (jpeg_kernels.hip): tables + per-block temporaries (one set, compute stream only), and per
    // staging parity the shared stream buffer, its page-locked landing buffer and the totals the device reports
    struct JpegState {
        int quality = 0, pairs = 0; // what the buffers below are sized for
        struct JpegTables *d_tab = nullptr;
        short *d_dc = nullptr;
        unsigned *d_bits = nullptr;
        unsigned long long *d_plane_bits = nullptr, *d_plane_base = nullptr;
        unsigned *d_stream[2] = {nullptr, nullptr};
        unsigned char *h_stream[2] = {nullptr, nullptr};
        size_t h_capacity[2] = {0, 0}; // page-locked landing buffers: sized to what batches actually need (ensure_jpeg_landing)
        unsigned long long *h_info[2] = {nullptr, nullptr}, *d_info[2] = {nullptr, nullptr}; // mapped page-locked
        unsigned long long *d_hdr = nullptr; // device copy of (total, overflow): the emit pass must not poll host memory
        size_t capacity = 0;
        std::vector<unsigned char> header;
    } jpeg;
    DfxHelper helper;              // host-side work beside the calling thread (dfx_helper.h): one thread per handle
    std::vector<int> h_slots;      // slot id of each new frame of the current batch
    std::vector<PairDesc> h_pairs; // descriptors of the current batch

    dfx_stats stats{};

    // Batches are numbered across calls: staging set, bounce buffer and event of batch q are those of parity q & 1.
    unsigned long long batch_seq = 0;
    std::vector<int> next_segments; // dfx_next_segments: clip lengths of the NEXT FlowBuffer (consumed by that call)
    // Deferred tails of dfx_submit_*: the last download of a FlowBuffer (and, for small frames, the hand-over from
    // the bounce buffer to the caller's buffers) completes on a helper thread while the next FlowBuffer is issued.
    // A tail stays registered in `tails` until its worker has FINISHED (done, set under tails_mtx): whoever asks about
    // a ticket or a staging parity — the collector's dfx_wait, the submitting thread's bounce-buffer guard, a
    // re-allocation — sees it and blocks on tails_cv until then.  Only finished tails are removed and joined.
    struct Tail {
        unsigned long long ticket = 0;
        int parity = 0;
        int rc = DFX_OK;
        std::string err;
        bool done = false;
        std::thread worker;
    };
    struct TailError { // a failed tail's status is kept until a dfx_wait that covers its ticket has reported it
        unsigned long long ticket;
        int rc;
        std::string err;
    };
    std::deque<std::unique_ptr<Tail>> tails;
    std::vector<TailError> tail_errors;
    std::mutex tails_mtx; // dfx_wait may be called from another thread than the one that submits
    std::condition_variable tails_cv;
    unsigned long long next_ticket = 1;
};

// Wait until the deferred tails with ticket <= up_to (0 = all) / of staging parity `parity` (-1 = any) have finished.
// report = true (dfx_wait): returns, and forgets, the first error of a tail with ticket <= up_to; report = false
// (housekeeping inside other entry points): errors stay recorded for the dfx_wait of their ticket.
int dfx_finish_tails(dfx_context *c, unsigned long long up_to, int parity, bool report = false);

#define HIPCHK(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            char buf_[512];                                                                                     \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,        \
                     __LINE__);                                                                                 \
            (ctx)->set_err(buf_);                                                                                 \
            return DFX_ERR_HIP;                                                                                 \
        }                                                                                                       \
    } while (0)

// Event flags of the hot path's events: with dfx_params.blocking_sync a hipEventSynchronize sleeps instead of spinning.
inline unsigned dfx_event_flags(const dfx_context *c, bool timing) {
    return (timing ? 0u : (unsigned)hipEventDisableTiming) | (c->prm.blocking_sync ? (unsigned)hipEventBlockingSync : 0u);
}
// hipStreamSynchronize spins (the device flags are whatever the process set up before this library saw the device);
// with blocking_sync the wait goes through a blocking event instead.  One waiter per context at a time (the calling
// thread of the calc / submit entry points).
inline hipError_t dfx_stream_wait(dfx_context *c, hipStream_t s) {
    if (!c->prm.blocking_sync || !c->ev_block)
        return hipStreamSynchronize(s);
    const hipError_t e = hipEventRecord(c->ev_block, s);
    return e != hipSuccess ? e : hipEventSynchronize(c->ev_block);
}

inline int dfx_fail(dfx_context *c, int code, const std::string &msg) {
    if (c)
        c->set_err(msg);
    return code;
}

inline int dfx_cv_round(double v) { return (int)std::lrint(v); } // round-half-even

template <class T> inline void dfx_free_dev(T *&p) {
    if (p) {
        (void)hipFree(p);
        p = nullptr;
    }
}
template <class T> inline void dfx_free_host(T *&p) {
    if (p) {
        (void)hipHostFree(p);
        p = nullptr;
    }
}

AlgoEngine *dfx_make_tvl1_engine(dfx_context *c);
AlgoEngine *dfx_make_farneback_engine(dfx_context *c);
AlgoEngine *dfx_make_brox_engine(dfx_context *c);
