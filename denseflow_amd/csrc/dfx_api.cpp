// dfx_api.cpp — the C ABI of include/dfx.h: argument checking, the FlowBuffer driver that is common
// to every algorithm, statistics and memory helpers.  The algorithms live in *_engine.cpp.
//
// Reference behaviour mirrored: DenseFlow::calc_optflows_imp, /root/reference/src/denseflow_gpu.cpp
// :282-370 — pair selection :315-316, per-pair upload/calc/download :317-339, M = max(N-|step|,0)
// flows per FlowBuffer :307-308.
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "dfx_internal.h"
#include "dfx_plan.h"
#include "jpeg_kernels.h"
#include "prepare_kernels.h"
#include "quantize_kernels.h"

namespace {

thread_local std::string g_create_error;

int fail_create(int code, const std::string &msg) {
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

} // namespace

int dfx_finish_tails(dfx_context *c, unsigned long long up_to, int parity, bool report) {
    auto matches = [&](const dfx_context::Tail &t) {
        return (up_to == 0 || t.ticket <= up_to) && (parity < 0 || t.parity == parity);
    };
    std::vector<std::unique_ptr<dfx_context::Tail>> finished;
    int rc = DFX_OK;
    {
        std::unique_lock<std::mutex> lock(c->tails_mtx);
        // a tail is visible to every caller until its worker is done: nobody can mistake "being joined" for "finished"
        c->tails_cv.wait(lock, [&] {
            for (const auto &t : c->tails)
                if (matches(*t) && !t->done)
                    return false;
            return true;
        });
        for (auto it = c->tails.begin(); it != c->tails.end();) {
            if (matches(**it)) {
                if ((*it)->rc != DFX_OK)
                    c->tail_errors.push_back({(*it)->ticket, (*it)->rc, (*it)->err});
                finished.push_back(std::move(*it));
                it = c->tails.erase(it);
            } else {
                ++it;
            }
        }
        if (report) {
            for (auto it = c->tail_errors.begin(); it != c->tail_errors.end();) {
                if (up_to == 0 || it->ticket <= up_to) {
                    if (rc == DFX_OK) {
                        rc = it->rc;
                        c->set_err(it->err);
                    }
                    it = c->tail_errors.erase(it);
                } else {
                    ++it;
                }
            }
        }
    }
    for (auto &t : finished) // done was the worker's last action: these joins return at once
        if (t->worker.joinable())
            t->worker.join();
    return rc;
}

namespace {

int ensure_img_staging(dfx_context *c, int img_need) {
    if (img_need > c->img_slots) {
        (void)dfx_finish_tails(c, 0, -1);
        HIPCHK(c, hipDeviceSynchronize());
        c->img_slots = 0; // a failed allocation below must not leave the old size standing
        for (auto &p : c->d_img) {
            dfx_free_dev(p);
            HIPCHK(c, hipMalloc(&p, (size_t)img_need * 2 * c->W * c->H)); // img_need x planes, then the y planes
        }
        c->img_slots = img_need;
    }
    return DFX_OK;
}

int ensure_png(dfx_context *c, int need) {
    if (need > c->png_slots) {
        (void)dfx_finish_tails(c, 0, -1);
        HIPCHK(c, hipDeviceSynchronize());
        c->png_slots = 0;
        dfx_free_dev(c->d_png_scratch);
        HIPCHK(c, hipMalloc(&c->d_png_scratch, quant_png_scratch_bytes(need)));
        for (int p = 0; p < 2; ++p) {
            dfx_free_host(c->h_png_bounds[p]);
            HIPCHK(c, hipHostMalloc((void **)&c->h_png_bounds[p], (size_t)need * 2 * sizeof(double), hipHostMallocMapped));
            HIPCHK(c, hipHostGetDevicePointer((void **)&c->d_png_bounds[p], c->h_png_bounds[p], 0));
        }
        c->png_slots = need;
    }
    return DFX_OK;
}

int ensure_src_staging(dfx_context *c, int need) {
    const size_t fb = c->in_row_bytes() * c->in_h();
    if (need > c->src_slots || fb != c->src_frame_bytes) {
        (void)dfx_finish_tails(c, 0, -1);
        HIPCHK(c, hipDeviceSynchronize());
        c->src_slots = 0;
        c->src_frame_bytes = 0;
        for (auto &p : c->d_src) {
            dfx_free_dev(p);
            HIPCHK(c, hipMalloc(&p, (size_t)need * fb));
        }
        c->src_slots = need;
        c->src_frame_bytes = fb;
    }
    return DFX_OK;
}

void free_jpeg(dfx_context *c) {
    auto &j = c->jpeg;
    dfx_free_dev(j.d_tab);
    dfx_free_dev(j.d_dc);
    dfx_free_dev(j.d_bits);
    dfx_free_dev(j.d_plane_bits);
    dfx_free_dev(j.d_plane_base);
    dfx_free_dev(j.d_hdr);
    for (int q = 0; q < 2; ++q) {
        dfx_free_dev(j.d_stream[q]);
        dfx_free_host(j.h_stream[q]);
        j.h_capacity[q] = 0;
        dfx_free_host(j.h_info[q]);
        j.d_info[q] = nullptr;
    }
    j.quality = j.pairs = 0;
    j.capacity = 0;
}

// Buffers of the device JPEG encoder for batches of up to `pairs` pairs at `quality`.  The shared stream buffer holds
// 4 bits per pixel on average over the batch (flow planes need ~0.5; a batch that does not fit is reported, not cut).
int ensure_jpeg(dfx_context *c, int pairs, int quality) {
    auto &j = c->jpeg;
    if (j.quality == quality && pairs <= j.pairs)
        return DFX_OK;
    (void)dfx_finish_tails(c, 0, -1);
    HIPCHK(c, hipDeviceSynchronize());
    free_jpeg(c);
    const size_t planes = 2 * (size_t)pairs, nblk = (size_t)((c->W + 7) / 8) * ((c->H + 7) / 8);
    JpegTables t;
    unsigned char q[64];
    jpeg_build_tables(quality, t, q);
    j.header = jpeg_file_header(c->W, c->H, q);
    HIPCHK(c, hipMalloc(&j.d_tab, sizeof(JpegTables)));
    HIPCHK(c, hipMemcpy(j.d_tab, &t, sizeof t, hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&j.d_dc, planes * nblk * sizeof(short)));
    HIPCHK(c, hipMalloc(&j.d_bits, planes * nblk * sizeof(unsigned)));
    HIPCHK(c, hipMalloc(&j.d_plane_bits, planes * 8));
    HIPCHK(c, hipMalloc(&j.d_plane_base, planes * 8));
    HIPCHK(c, hipMalloc(&j.d_hdr, 16));
    j.capacity = ((planes * (size_t)c->W * c->H / 2 + (64u << 10)) + 255) & ~(size_t)255;
    for (int p = 0; p < 2; ++p) {
        HIPCHK(c, hipMalloc(&j.d_stream[p], j.capacity));
        // The page-locked landing buffer starts at 1 bit per pixel (flow planes code to ~0.3-0.5) and grows to what a batch
        // really needs (ensure_jpeg_landing): pinning 4 bits per pixel twice was ~0.1 s of a 1080p handle's first call
        // (profiles/round5/e2e/) for bytes that never arrive.
        j.h_capacity[p] = (j.capacity / 4 + 255) & ~(size_t)255;
        HIPCHK(c, hipHostMalloc(&j.h_stream[p], j.h_capacity[p], hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc(&j.h_info[p], (2 + 2 * planes) * 8, hipHostMallocMapped));
        std::memset(j.h_info[p], 0, (2 + 2 * planes) * 8);
        HIPCHK(c, hipHostGetDevicePointer((void **)&j.d_info[p], j.h_info[p], 0));
    }
    j.quality = quality;
    j.pairs = pairs;
    return DFX_OK;
}

// The shared stream buffers (device + page-locked landing buffer, both parities) re-sized to hold `need` bytes: called
// when the scan pass of a batch has measured more than the 4 bits per pixel ensure_jpeg() provides for (noise, film
// grain), so that the batch can be coded again from the planes that are still on the device — the flows are not
// recomputed and the caller does not have to fall back to the host encoder.
int grow_jpeg_streams(dfx_context *c, unsigned long long need) {
    auto &j = c->jpeg;
    (void)dfx_finish_tails(c, 0, -1); // a deferred tail may still be reading a landing buffer
    HIPCHK(c, hipDeviceSynchronize());
    const size_t cap = (((size_t)need + (size_t)need / 4 + (64u << 10)) + 255) & ~(size_t)255;
    // All four new buffers first, swapped in only when every allocation has succeeded: a failure part-way leaves the
    // encoder exactly as it was (old buffers, old capacity) and the call reports the error.
    unsigned *nd[2] = {nullptr, nullptr};
    unsigned char *nh[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int p = 0; p < 2 && e == hipSuccess; ++p) {
        e = hipMalloc((void **)&nd[p], cap);
        if (e == hipSuccess)
            e = hipHostMalloc((void **)&nh[p], cap, hipHostMallocDefault);
    }
    if (e != hipSuccess) {
        for (int p = 0; p < 2; ++p) {
            if (nd[p])
                (void)hipFree(nd[p]);
            if (nh[p])
                (void)hipHostFree(nh[p]);
        }
        (void)hipGetLastError();
        return dfx_fail(c, DFX_ERR_HIP, "growing the JPEG stream buffers failed; the encoder keeps its old buffers");
    }
    for (int p = 0; p < 2; ++p) {
        dfx_free_dev(j.d_stream[p]);
        dfx_free_host(j.h_stream[p]);
        j.d_stream[p] = nd[p];
        j.h_stream[p] = nh[p];
        j.h_capacity[p] = cap;
    }
    j.capacity = cap;
    return DFX_OK;
}

// The landing buffer of parity q holds at least `need` bytes (the total a batch's scan pass measured).  Nothing reads or
// writes h_stream[q] when this is called: the caller has finished the tails of that parity and its copy is not enqueued yet.
int ensure_jpeg_landing(dfx_context *c, int q, size_t need) {
    auto &j = c->jpeg;
    if (need <= j.h_capacity[q])
        return DFX_OK;
    const size_t cap = std::min(j.capacity, ((need + need / 2) + 255) & ~(size_t)255);
    unsigned char *nh = nullptr;
    if (hipHostMalloc((void **)&nh, cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return dfx_fail(c, DFX_ERR_HIP, "growing the JPEG landing buffer failed");
    }
    dfx_free_host(j.h_stream[q]);
    j.h_stream[q] = nh;
    j.h_capacity[q] = cap;
    return DFX_OK;
}

int ensure_bounce(dfx_context *c, size_t in_bytes, size_t out_bytes) {
    if (in_bytes > c->h_in_bytes) {
        (void)dfx_finish_tails(c, 0, -1);
        HIPCHK(c, hipDeviceSynchronize());
        c->h_in_bytes = 0;
        for (auto &p : c->h_in) {
            dfx_free_host(p);
            HIPCHK(c, hipHostMalloc(&p, in_bytes, hipHostMallocDefault));
        }
        c->h_in_bytes = in_bytes;
    }
    if (out_bytes > c->h_out_bytes) {
        (void)dfx_finish_tails(c, 0, -1);
        HIPCHK(c, hipDeviceSynchronize());
        c->h_out_bytes = 0;
        for (auto &p : c->h_out) {
            dfx_free_host(p);
            HIPCHK(c, hipHostMalloc(&p, out_bytes, hipHostMallocDefault));
        }
        c->h_out_bytes = out_bytes;
    }
    return DFX_OK;
}

int ensure_staging(dfx_context *c, int u8_need, int flow_need) {
    if (u8_need > c->u8_slots) {
        (void)dfx_finish_tails(c, 0, -1);
        HIPCHK(c, hipDeviceSynchronize());
        c->u8_slots = 0;
        for (auto &p : c->d_u8) {
            dfx_free_dev(p);
            HIPCHK(c, hipMalloc(&p, (size_t)u8_need * c->W * c->H));
        }
        c->u8_slots = u8_need;
    }
    if (flow_need > c->flow_slots) {
        (void)dfx_finish_tails(c, 0, -1);
        HIPCHK(c, hipDeviceSynchronize());
        c->flow_slots = 0;
        for (auto &p : c->d_flow_out) {
            dfx_free_dev(p);
            HIPCHK(c, hipMalloc(&p, (size_t)flow_need * c->W * c->H * 2 * sizeof(float)));
        }
        c->flow_slots = flow_need;
    }
    return DFX_OK;
}

// Where the flows of a FlowBuffer go: float (u, v) fields or planes bounded to 8 bits on the device.
struct OutSpec {
    bool quantized = false;
    double lo = 0, hi = 0;
    // float output
    float *const *flows = nullptr; // host mode: one pointer per flow, out_pitch bytes per row
    size_t out_pitch = 0;
    float *d_flows = nullptr; // device mode: flow i dense at d_flows + i*d_flow_stride
    size_t d_flow_stride = 0;
    // 8-bit output
    uint8_t *const *img_x = nullptr, *const *img_y = nullptr; // host mode: one pointer per plane
    size_t img_pitch = 0;                                     // bytes per row (host and device mode)
    uint8_t *d_img_x = nullptr, *d_img_y = nullptr;           // device mode: plane i at + i*d_img_stride
    size_t d_img_stride = 0;
    // the -st=png scheme (implies quantized; lo / hi unused): planes scaled by the reference's per-flow adaptive bounds,
    // which go to bounds[2 * i] = {bound_x, bound_y} (host mode: filled when the call returns) or d_bounds (device mode)
    bool png = false;
    double *bounds = nullptr, *d_bounds = nullptr;
    // JPEG output (host mode; implies quantized): one file per plane into jpg_x[i] / jpg_y[i] (jpg_capacity bytes each)
    bool jpeg = false;
    int quality = 95;
    uint8_t *const *jpg_x = nullptr, *const *jpg_y = nullptr;
    size_t jpg_capacity = 0;
    uint32_t *size_x = nullptr, *size_y = nullptr;
};

// One frame / plane between host and device.  Dense rows (pitch == row bytes on both sides) go as ONE linear copy:
// a 2-D copy of a small frame costs several times the linear one, and a FlowBuffer of 224x224 frames is hundreds
// of them.
inline hipError_t copy_rows_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t row_bytes,
                                  size_t rows, hipMemcpyKind kind, hipStream_t s) {
    if (dpitch == row_bytes && spitch == row_bytes)
        return hipMemcpyAsync(dst, src, row_bytes * rows, kind, s);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, kind, s);
}


// Shared driver for host- and device-resident frames.
//   host mode  : frames[i] host pointers (frame_pitch), flows[i] host pointers (out_pitch bytes).
//                Copies run on their own stream through two staging sets: the frames of batch i+1 go up
//                and the flows of batch i-1 come down while batch i computes (the reference uploads,
//                computes and downloads one pair at a time with a blocking download, :317-339).
//   device mode: d_frames / d_flows contiguous device arrays, no copies at all.
//   ticket != nullptr (dfx_submit_*): the call returns when the device work of the FlowBuffer is done and every
//                batch but the last has been handed over; the last download (+ hand-over) finishes on a helper
//                thread (dfx_context::Tail) and is awaited by dfx_wait(ticket).
int calc_batch_body(dfx_context *c, const uint8_t *const *frames, size_t frame_pitch, const uint8_t *d_frames,
                    size_t d_pitch, size_t d_frame_stride, int n_frames, int step, const OutSpec &out,
                    unsigned long long *ticket) {
    // dfx_next_segments applies to this call only, whatever becomes of it
    std::vector<int> seg;
    seg.swap(c->next_segments);
    if (ticket)
        *ticket = 0;
    else
        (void)dfx_finish_tails(c, 0, -1); // synchronous entry points never run beside a deferred tail
    if (n_frames < 0 || step == 0 || step < -(1 << 30) || step > (1 << 30)) // (|INT_MIN| is not an int)
        return dfx_fail(c, DFX_ERR_INVALID, "n_frames must be >= 0 and step non-zero");
    const int astep = std::abs(step);
    // The FlowBuffer's pairs as (frame a, frame b), frame ids counted over the whole buffer.  One clip: pair i is
    // (i, i + step) for step > 0, (i - step, i) otherwise, M = max(N - |step|, 0) of them (src/denseflow_gpu.cpp:307-316).
    // Several clips joined (dfx_next_segments): the same rule inside every clip, no pair across a clip boundary.
    if (seg.empty())
        seg.push_back(n_frames);
    {
        long long total = 0;
        for (int n : seg) {
            if (n < 0)
                return dfx_fail(c, DFX_ERR_INVALID, "dfx_next_segments: negative clip length");
            total += n;
        }
        if (total != n_frames)
            return dfx_fail(c, DFX_ERR_INVALID, "dfx_next_segments: the clip lengths do not add up to n_frames");
    }
    const DfxPairs pairs = dfx_build_pairs(seg, step); // dfx_plan.h: pure host logic, CPU-tested
    const int M = pairs.size();
    if (M == 0)
        return DFX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    AlgoEngine *E = c->engine;
    int B = E->batch();
    const int F_need = std::max(dfx_frames_needed(pairs, B), std::min(B, M) + astep);
    int rc = E->ensure_frame_slots(F_need);
    if (rc != DFX_OK)
        return rc;
    const bool host_mode = frames != nullptr;
    // float flows land in the caller's device array, or in a staging set when they are copied to the host
    // or only feed the bounding kernel
    const bool prep = c->prepares(); // inputs are source-format frames: convert / resize them on the device first
    rc = ensure_staging(c, (host_mode || prep) ? F_need : 0, (host_mode || out.quantized) ? B : 0);
    if (rc != DFX_OK)
        return rc;
    if (prep && host_mode) {
        rc = ensure_src_staging(c, F_need);
        if (rc != DFX_OK)
            return rc;
    }
    if (out.quantized && host_mode) {
        rc = ensure_img_staging(c, B);
        if (rc != DFX_OK)
            return rc;
    }
    if (out.jpeg) {
        rc = ensure_jpeg(c, B, out.quality);
        if (rc != DFX_OK)
            return rc;
    }
    if (out.png) {
        rc = ensure_png(c, B);
        if (rc != DFX_OK)
            return rc;
    }
    const size_t plane = (size_t)c->W * c->H;
    // Small frames: an asynchronous copy costs ~10 us of driver time whatever its size, and a 300-frame clip of
    // 224x224 frames is ~900 of them (a third of the batch's compute time).  Such FlowBuffers go through page-locked
    // bounce buffers instead: the host gathers / scatters the frames with memcpy and the copy stream moves one block
    // per batch and direction.
    const size_t in_fb = c->in_row_bytes() * c->in_h();
    const size_t out_pb = out.quantized ? 2 * plane : plane * 8; // bytes per pair leaving the device
    // Decided per direction: a 224x224 frame is 50 KB (gathered), but its float flow is 401 KB — one direct copy per
    // flow (~10 us of driver time) is cheaper than a second pass of host memcpy over 120 MB per clip.
    const bool bounce_in = host_mode && in_fb <= (256u << 10) && (size_t)F_need * in_fb <= (256u << 20);
    const bool bounce = !out.jpeg && bounce_in && out_pb <= (256u << 10) && (size_t)B * out_pb <= (256u << 20); // results
    if (bounce_in) {
        rc = ensure_bounce(c, (size_t)F_need * in_fb, bounce ? (size_t)B * out_pb : 0);
        if (rc != DFX_OK)
            return rc;
    } else if (host_mode && M <= B && M >= 32) {
        // Large frames, and the whole FlowBuffer would be one batch: nothing could overlap its copies.  Two balanced
        // batches put the second upload and the first download under the compute (the engine's batch is sized for
        // the device-resident path, where a bigger batch is simply better: 336 / 362 pairs/s at 32 / 128 for TVL1).
        B = (M + 1) / 2;
    }
    const int F = E->frame_slots();
    c->h_slots.resize(F);
    c->h_pairs.resize(B);

    // Frames [lo of its first pair, hi of its last pair] must be resident for a batch; earlier batches already prepared
    // the ids below their own end.  Frame id f lives in slot f % F; F >= that range, so a batch never evicts what it needs.
    const std::vector<DfxBatchPlan> plan = dfx_plan_batches(pairs, B);
    // batches are numbered across calls (q = seq0 + k): batch q uses staging set / bounce buffer / events q & 1
    const unsigned long long seq0 = c->batch_seq;
    c->batch_seq += plan.size();
    auto par = [&](size_t k) -> int { return (int)((seq0 + k) & 1ull); };
    auto upload = [&](size_t k) -> int { // host frames of batch k -> staging set par(k) (upload stream)
        const DfxBatchPlan &p = plan[k];
        const size_t rb = c->in_row_bytes(), fb = rb * c->in_h();
        unsigned char *dst = prep ? c->d_src[par(k)] : c->d_u8[par(k)];
        // the staging set was last read by the frame preparation of batch q-2 (compute stream)
        if (seq0 + k >= 2)
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_compute[par(k)], 0));
        if (bounce_in) {
            if (seq0 + k >= 2) // the copy that last read this bounce buffer (batch q-2) has long finished; make it formal
                HIPCHK(c, hipEventSynchronize(c->ev_h2d[par(k)]));
            unsigned char *hb = c->h_in[par(k)];
            for (int j = 0; j < p.n_new; ++j) {
                const uint8_t *src = frames[p.first_new + j];
                if (frame_pitch == rb)
                    std::memcpy(hb + (size_t)j * fb, src, fb);
                else
                    for (int y = 0; y < c->in_h(); ++y)
                        std::memcpy(hb + (size_t)j * fb + (size_t)y * rb, src + (size_t)y * frame_pitch, rb);
            }
            if (p.n_new > 0)
                HIPCHK(c, hipMemcpyAsync(dst, hb, (size_t)p.n_new * fb, hipMemcpyHostToDevice, c->copy_stream));
        } else {
            for (int j = 0; j < p.n_new; ++j)
                HIPCHK(c, copy_rows_async(dst + (size_t)j * fb, rb, frames[p.first_new + j], frame_pitch, rb, c->in_h(),
                                          hipMemcpyHostToDevice, c->copy_stream));
        }
        HIPCHK(c, hipEventRecord(c->ev_h2d[par(k)], c->copy_stream));
        return DFX_OK;
    };
    // JPEG mode: what the device reported for each batch (read after the batch's stream synchronisation)
    struct JpegBatch {
        unsigned long long total = 0, overflow = 0;
        std::vector<unsigned long long> bits, base; // per plane: x planes of the batch, then its y planes
    };
    std::vector<JpegBatch> jb(out.jpeg ? plan.size() : 0);
    auto download = [&](size_t k) -> int { // flows of batch k: staging set par(k) -> host (download stream)
        const DfxBatchPlan &p = plan[k];
        const int q = par(k);
        HIPCHK(c, hipStreamWaitEvent(c->d2h_stream, c->ev_compute[q], 0));
        if (out.jpeg) { // the batch's entropy-coded segments, one block; assemble_jpeg(k) turns them into files later
            const int trc = dfx_finish_tails(c, 0, q); // a deferred tail may still be reading this landing buffer
            if (trc != DFX_OK)
                return trc;
            if (jb[k].overflow)
                return dfx_fail(c, DFX_ERR_UNSUPPORTED,
                                "JPEG: the batch's streams do not fit the stream buffer (use the 8-bit plane output and encode "
                                "on the host)");
            if (jb[k].total > 0) {
                const int grc = ensure_jpeg_landing(c, q, (size_t)jb[k].total);
                if (grc != DFX_OK)
                    return grc;
                HIPCHK(c, hipMemcpyAsync(c->jpeg.h_stream[q], c->jpeg.d_stream[q], (size_t)jb[k].total,
                                         hipMemcpyDeviceToHost, c->d2h_stream));
            }
            HIPCHK(c, hipEventRecord(c->ev_d2h[q], c->d2h_stream));
            return DFX_OK;
        }
        if (bounce) { // one block per plane kind; scatter(k) hands the rows to the caller's buffers later
            // a deferred tail of an earlier FlowBuffer may still have to empty this bounce buffer
            const int trc = dfx_finish_tails(c, 0, q);
            if (trc != DFX_OK)
                return trc;
            unsigned char *hb = c->h_out[q];
            if (out.quantized) {
                HIPCHK(c, hipMemcpyAsync(hb, c->d_img[q], (size_t)p.nb * plane, hipMemcpyDeviceToHost, c->d2h_stream));
                HIPCHK(c, hipMemcpyAsync(hb + (size_t)p.nb * plane, c->d_img[q] + (size_t)c->img_slots * plane,
                                         (size_t)p.nb * plane, hipMemcpyDeviceToHost, c->d2h_stream));
            } else {
                HIPCHK(c, hipMemcpyAsync(hb, c->d_flow_out[q], (size_t)p.nb * plane * 8, hipMemcpyDeviceToHost,
                                         c->d2h_stream));
            }
            HIPCHK(c, hipEventRecord(c->ev_d2h[q], c->d2h_stream));
            return DFX_OK;
        }
        for (int j = 0; j < p.nb; ++j) {
            if (out.quantized) {
                const unsigned char *sx = c->d_img[q] + (size_t)j * plane;
                const unsigned char *sy = c->d_img[q] + ((size_t)c->img_slots + j) * plane;
                HIPCHK(c, copy_rows_async(out.img_x[p.i0 + j], out.img_pitch, sx, c->W, c->W, c->H,
                                          hipMemcpyDeviceToHost, c->d2h_stream));
                HIPCHK(c, copy_rows_async(out.img_y[p.i0 + j], out.img_pitch, sy, c->W, c->W, c->H,
                                          hipMemcpyDeviceToHost, c->d2h_stream));
            } else {
                HIPCHK(c, copy_rows_async(out.flows[p.i0 + j], out.out_pitch, c->d_flow_out[q] + (size_t)j * plane * 2,
                                          (size_t)c->W * 8, (size_t)c->W * 8, c->H, hipMemcpyDeviceToHost,
                                          c->d2h_stream));
            }
        }
        HIPCHK(c, hipEventRecord(c->ev_d2h[q], c->d2h_stream));
        return DFX_OK;
    };

    auto copy_rows = [](void *dst, size_t dpitch, const void *src, size_t row_bytes, int rows) {
        if (dpitch == row_bytes)
            std::memcpy(dst, src, row_bytes * rows);
        else
            for (int y = 0; y < rows; ++y)
                std::memcpy((char *)dst + (size_t)y * dpitch, (const char *)src + (size_t)y * row_bytes, row_bytes);
    };
    auto scatter = [&](size_t k) -> int { // bounce mode: results of batch k -> the caller's buffers (host memcpy)
        const DfxBatchPlan &p = plan[k];
        HIPCHK(c, hipEventSynchronize(c->ev_d2h[par(k)]));
        if (out.jpeg) { // header + byte-stuffed segment + EOI for every plane of the batch
            const unsigned char *hb = c->jpeg.h_stream[par(k)];
            for (int j = 0; j < 2 * p.nb; ++j) {
                const bool is_y = j >= p.nb;
                const int i = p.i0 + (is_y ? j - p.nb : j);
                const size_t n = jpeg_assemble(c->jpeg.header, hb + jb[k].base[j], jb[k].bits[j],
                                               is_y ? out.jpg_y[i] : out.jpg_x[i], out.jpg_capacity);
                if (n == 0)
                    return dfx_fail(c, DFX_ERR_UNSUPPORTED,
                                    "JPEG: jpg_capacity is too small for an encoded plane (encode this FlowBuffer's 8-bit planes "
                                    "on the host)");
                (is_y ? out.size_y : out.size_x)[i] = (uint32_t)n;
            }
            return DFX_OK;
        }
        const unsigned char *hb = c->h_out[par(k)];
        for (int j = 0; j < p.nb; ++j) {
            if (out.quantized) {
                copy_rows(out.img_x[p.i0 + j], out.img_pitch, hb + (size_t)j * plane, c->W, c->H);
                copy_rows(out.img_y[p.i0 + j], out.img_pitch, hb + ((size_t)p.nb + j) * plane, c->W, c->H);
            } else {
                copy_rows(out.flows[p.i0 + j], out.out_pitch, hb + (size_t)j * plane * 8, (size_t)c->W * 8, c->H);
            }
        }
        return DFX_OK;
    };

    if (host_mode) {
        rc = upload(0);
        if (rc != DFX_OK)
            return rc;
    }
    // the handle's helper thread runs the host-side post-processing of the previous batch; its job captures this
    // function's locals by reference, so it is finished on every path out of here
    struct PostGuard {
        DfxHelper &h;
        void start(std::function<int()> fn) { h.start(std::move(fn)); }
        int finish() { return h.finish(); }
        ~PostGuard() { (void)h.finish(); }
    } post{c->helper};
    for (size_t k = 0; k < plan.size(); ++k) {
        const DfxBatchPlan &p = plan[k];
        if (host_mode) {
            // flows of batch k-1 down (download stream, after its compute), frames of batch k+1 up (upload stream,
            // after the frame preparation of batch k-1, which last read that staging set)
            if (k >= 1) {
                rc = download(k - 1);
                if (rc != DFX_OK)
                    return rc;
            }
            // Host work that can run beside this thread driving batch k (the TVL1 engine polls the device inside
            // run_pairs), on a helper thread: hand the results of batch k-1 over (rows to the caller's buffers / JPEG
            // files assembled; its download was enqueued just above and is a fraction of a batch's compute time), and — for
            // small frames — gather the frames of batch k+1 into the page-locked bounce buffer and send them up: 2048
            // frames of 224 x 224 are 100 MB of host memcpy, 6 % of their batch's compute time when the GPU waits for it.
            const bool hand_over = (bounce || out.jpeg) && k >= 1;
            const bool up_next = k + 1 < plan.size();
            if (up_next && !bounce_in) { // large frames: a few asynchronous copies to enqueue, nothing to gather
                rc = upload(k + 1);
                if (rc != DFX_OK)
                    return rc;
            }
            if (hand_over || (up_next && bounce_in)) {
                post.start([&, k, hand_over, up_next] {
                    (void)hipSetDevice(c->device);
                    int r = hand_over ? scatter(k - 1) : DFX_OK;
                    if (r == DFX_OK && up_next && bounce_in)
                        r = upload(k + 1);
                    return r;
                });
            }
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_h2d[par(k)], 0));
            if (seq0 + k >= 2) // flow staging set par(k) must have been drained by the download of batch q-2
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_d2h[par(k)], 0));
        }
        HIPCHK(c, hipEventRecord(c->ev_t0, c->stream));
        for (int j = 0; j < p.n_new; ++j)
            c->h_slots[j] = (int)((p.first_new + j) % F);
        if (p.n_new > 0) {
            if (prep) { // cvtColor + cv::resize of load_frames_batch (src/denseflow_gpu.cpp:163, :169), on the device
                if (host_mode)
                    prepare_launch(c->stream, c->d_src[par(k)], (long long)c->in_row_bytes(),
                                   (long long)c->src_frame_bytes, c->src_w, c->src_h, c->src_ch, p.n_new,
                                   c->d_u8[par(k)], c->W, (long long)plane, c->W, c->H);
                else
                    prepare_launch(c->stream, d_frames + (size_t)p.first_new * d_frame_stride, (long long)d_pitch,
                                   (long long)d_frame_stride, c->src_w, c->src_h, c->src_ch, p.n_new, c->d_u8[par(k)],
                                   c->W, (long long)plane, c->W, c->H);
                HIPCHK(c, hipGetLastError());
                c->stats.kernel_launches += 1;
            }
            if (host_mode || prep)
                rc = E->build_frames(c->d_u8[par(k)], (long long)c->W * c->H, c->W, p.n_new, c->h_slots.data());
            else
                rc = E->build_frames(d_frames + (size_t)p.first_new * d_frame_stride, (long long)d_frame_stride,
                                     (long long)d_pitch, p.n_new, c->h_slots.data());
            if (rc != DFX_OK)
                return rc;
        }
        // pair i of a clip: a = (step>0 ? i : i-step), b = (step>0 ? i+step : i)   (src/denseflow_gpu.cpp:315-316)
        for (int j = 0; j < p.nb; ++j) {
            const int i = p.i0 + j;
            c->h_pairs[j].frame_a = dfx_pair_a(pairs, i, step) % F;
            c->h_pairs[j].frame_b = dfx_pair_b(pairs, i, step) % F;
        }
        const bool staged = host_mode || out.quantized;
        float *dst = staged ? c->d_flow_out[par(k)] : out.d_flows + (size_t)p.i0 * out.d_flow_stride;
        const long long dst_stride = staged ? (long long)plane * 2 : (long long)out.d_flow_stride;
        rc = E->run_pairs(p.nb, c->h_pairs.data(), dst, dst_stride);
        if (rc != DFX_OK)
            return rc;
        if (out.png) { // convertFlowToPngImage's bounds and planes on the device (src/common.cpp:18-46)
            if (host_mode)
                quant_launch_flow_to_png_planes(c->stream, dst, dst_stride, p.nb, c->W, c->H, c->d_png_scratch,
                                                c->d_png_bounds[par(k)], c->d_img[par(k)],
                                                c->d_img[par(k)] + (size_t)c->img_slots * plane, c->W, (long long)plane);
            else
                quant_launch_flow_to_png_planes(c->stream, dst, dst_stride, p.nb, c->W, c->H, c->d_png_scratch,
                                                out.d_bounds + 2 * (size_t)p.i0,
                                                out.d_img_x + (size_t)p.i0 * out.d_img_stride,
                                                out.d_img_y + (size_t)p.i0 * out.d_img_stride, (long long)out.img_pitch,
                                                (long long)out.d_img_stride);
            HIPCHK(c, hipGetLastError());
            c->stats.kernel_launches += 4;
        } else if (out.quantized) { // convertFlowToImage on the device (src/common.cpp:4-16)
            if (host_mode)
                quant_launch_flow_to_u8(c->stream, dst, dst_stride, p.nb, c->W, c->H, out.lo, out.hi, c->d_img[par(k)],
                                        c->d_img[par(k)] + (size_t)c->img_slots * plane, c->W, (long long)plane);
            else
                quant_launch_flow_to_u8(c->stream, dst, dst_stride, p.nb, c->W, c->H, out.lo, out.hi,
                                        out.d_img_x + (size_t)p.i0 * out.d_img_stride,
                                        out.d_img_y + (size_t)p.i0 * out.d_img_stride, (long long)out.img_pitch,
                                        (long long)out.d_img_stride);
            HIPCHK(c, hipGetLastError());
            c->stats.kernel_launches += 1;
        }
        auto launch_jpeg = [&]() -> int { // imencode(".jpg") of both planes of every flow, on the device (src/common.cpp:56-57)
            JpegCtx jc;
            jc.planes = c->d_img[par(k)];
            jc.plane_stride = (long long)plane;
            jc.pitch = c->W, jc.w = c->W, jc.h = c->H, jc.bw = (c->W + 7) / 8, jc.bh = (c->H + 7) / 8;
            jc.n_planes = 2 * p.nb, jc.n_x = p.nb, jc.y_first = c->img_slots;
            jc.tab = c->jpeg.d_tab, jc.dc = c->jpeg.d_dc, jc.bits = c->jpeg.d_bits;
            jc.plane_bits = c->jpeg.d_plane_bits, jc.plane_base = c->jpeg.d_plane_base;
            jc.stream = c->jpeg.d_stream[par(k)], jc.capacity_bytes = c->jpeg.capacity;
            jc.info = c->jpeg.d_info[par(k)], jc.hdr = c->jpeg.d_hdr;
            jpeg_launch_encode(c->stream, jc);
            HIPCHK(c, hipGetLastError());
            c->stats.kernel_launches += 5;
            return DFX_OK;
        };
        if (out.jpeg) {
            rc = launch_jpeg();
            if (rc != DFX_OK)
                return rc;
        }
        HIPCHK(c, hipEventRecord(c->ev_t1, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_compute[par(k)], c->stream));
        HIPCHK(c, dfx_stream_wait(c, c->stream)); // the engines' statistics read-backs are complete
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
        c->stats.device_ms += ms;
        rc = E->account(p.nb);
        if (rc != DFX_OK)
            return rc;
        rc = post.finish(); // batch k-1 is in the caller's buffers
        if (rc != DFX_OK)
            return rc;
        if (out.png && host_mode) // the stream is idle: this batch's bounds are in the mapped block
            std::memcpy(out.bounds + 2 * (size_t)p.i0, c->h_png_bounds[par(k)], (size_t)p.nb * 2 * sizeof(double));
        if (out.jpeg) { // the stream is idle: the totals of this batch are in the mapped block
            const unsigned long long *hi = c->jpeg.h_info[par(k)];
            if (hi[1] != 0) {
                // The batch's streams do not fit the shared buffer (sized for 4 bits per pixel): nothing was written, but
                // the scan pass has measured what they need.  Grow the buffers to that and code the batch again — its
                // bounded planes are still in staging set par(k), the flows are not recomputed.
                rc = grow_jpeg_streams(c, hi[0]);
                if (rc != DFX_OK)
                    return rc;
                rc = launch_jpeg();
                if (rc != DFX_OK)
                    return rc;
                HIPCHK(c, hipEventRecord(c->ev_compute[par(k)], c->stream));
                HIPCHK(c, dfx_stream_wait(c, c->stream));
            }
            jb[k].total = hi[0], jb[k].overflow = hi[1];
            for (int j = 0; j < 2 * p.nb; ++j) {
                jb[k].bits.push_back(hi[2 + 2 * j]);
                jb[k].base.push_back(hi[2 + 2 * j + 1]);
            }
        }
    }
    if (host_mode) {
        const size_t last = plan.size() - 1;
        rc = download(last);
        if (rc != DFX_OK)
            return rc;
        // A plane of the last batch that could exceed jpg_capacity if every byte of it had to be stuffed (above 4 bits per
        // pixel against dfx_jpeg_capacity: noise) is assembled HERE, synchronously: "does not fit" is then DFX_ERR_UNSUPPORTED
        // from this call — the status that means "encode this FlowBuffer on the host" — and never an error of a deferred
        // tail, which the host shell could only treat as fatal.
        bool may_not_fit = false;
        for (size_t j = 0; out.jpeg && j < jb[last].bits.size(); ++j)
            may_not_fit = may_not_fit || c->jpeg.header.size() + 2 * (size_t)((jb[last].bits[j] >> 3) + 1) + 2 > out.jpg_capacity;
        if (ticket && !may_not_fit) {
            // deferred tail: wait for the last download and hand its rows over on a helper thread, so that the
            // caller can issue the next FlowBuffer now (its uploads run on the other copy stream)
            std::unique_ptr<dfx_context::Tail> t(new dfx_context::Tail());
            t->ticket = c->next_ticket++;
            t->parity = par(last);
            dfx_context::Tail *tp = t.get();
            const DfxBatchPlan lp = plan[last];
            hipEvent_t ev = c->ev_d2h[par(last)];
            const unsigned char *hb = bounce ? c->h_out[par(last)] : nullptr;
            const int W = c->W, H = c->H, dev = c->device;
            // the caller's pointer arrays need not outlive the submit call: copy the last batch's entries
            std::vector<void *> dst_a, dst_b;
            for (int j = 0; bounce && j < lp.nb; ++j) {
                if (out.quantized) {
                    dst_a.push_back(out.img_x[lp.i0 + j]);
                    dst_b.push_back(out.img_y[lp.i0 + j]);
                } else {
                    dst_a.push_back(out.flows[lp.i0 + j]);
                }
            }
            const bool quant = out.quantized;
            const size_t dpitch = quant ? out.img_pitch : out.out_pitch;
            // JPEG mode: the last batch's planes are assembled by the tail (header + stuffing is host work)
            const bool jpeg = out.jpeg;
            const unsigned char *jhb = jpeg ? c->jpeg.h_stream[par(last)] : nullptr;
            const std::vector<unsigned char> jheader = jpeg ? c->jpeg.header : std::vector<unsigned char>();
            const JpegBatch jlast = jpeg ? jb[last] : JpegBatch();
            std::vector<unsigned char *> jdst;
            std::vector<uint32_t *> jsize;
            const size_t jcap = out.jpg_capacity;
            for (int j = 0; jpeg && j < 2 * lp.nb; ++j) {
                const bool is_y = j >= lp.nb;
                const int i = lp.i0 + (is_y ? j - lp.nb : j);
                jdst.push_back(is_y ? out.jpg_y[i] : out.jpg_x[i]);
                jsize.push_back((is_y ? out.size_y : out.size_x) + i);
            }
            std::mutex *mtx = &c->tails_mtx;
            std::condition_variable *cv = &c->tails_cv;
            t->worker = std::thread([=]() {
                (void)hipSetDevice(dev);
                const hipError_t e = hipEventSynchronize(ev);
                int wrc = DFX_OK;
                std::string werr;
                if (e != hipSuccess) {
                    wrc = DFX_ERR_HIP;
                    werr = std::string("deferred download failed: ") + hipGetErrorString(e);
                } else if (jpeg) {
                    for (size_t j = 0; j < jdst.size(); ++j) {
                        const size_t n = jpeg_assemble(jheader, jhb + jlast.base[j], jlast.bits[j], jdst[j], jcap);
                        if (n == 0 && wrc == DFX_OK) {
                            wrc = DFX_ERR_INVALID;
                            werr = "JPEG: jpg_capacity is too small for an encoded plane";
                        }
                        *jsize[j] = (uint32_t)n;
                    }
                } else if (hb) {
                    const size_t pl = (size_t)W * H;
                    for (int j = 0; j < lp.nb; ++j) {
                        if (quant) {
                            copy_rows(dst_a[j], dpitch, hb + (size_t)j * pl, (size_t)W, H);
                            copy_rows(dst_b[j], dpitch, hb + ((size_t)lp.nb + j) * pl, (size_t)W, H);
                        } else {
                            copy_rows(dst_a[j], dpitch, hb + (size_t)j * pl * 8, (size_t)W * 8, H);
                        }
                    }
                }
                {   // last action: publish the result; the context outlives every tail (dfx_destroy drains them)
                    std::lock_guard<std::mutex> lock(*mtx);
                    tp->rc = wrc;
                    tp->err = werr;
                    tp->done = true;
                }
                cv->notify_all();
            });
            *ticket = t->ticket;
            {
                std::lock_guard<std::mutex> lock(c->tails_mtx);
                c->tails.push_back(std::move(t));
            }
            return DFX_OK;
        }
        HIPCHK(c, dfx_stream_wait(c, c->d2h_stream));
        if (bounce || out.jpeg) {
            rc = scatter(last);
            if (rc != DFX_OK)
                return rc;
        }
    }
    return DFX_OK;
}

// An error return may leave asynchronous copies in flight that target caller-owned (often pool-recycled) buffers:
// drain every stream and every deferred tail before handing the error back.
int calc_batch_impl(dfx_context *c, const uint8_t *const *frames, size_t frame_pitch, const uint8_t *d_frames,
                    size_t d_pitch, size_t d_frame_stride, int n_frames, int step, const OutSpec &out,
                    unsigned long long *ticket = nullptr) {
    const int rc = calc_batch_body(c, frames, frame_pitch, d_frames, d_pitch, d_frame_stride, n_frames, step, out, ticket);
    if (rc != DFX_OK) {
        const std::string keep = c->get_err();
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->d2h_stream);
        (void)dfx_finish_tails(c, 0, -1);
        c->set_err(keep);
    }
    return rc;
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
        if (out)
            *out = DFX_ALGO_TVL1;
        return DFX_OK;
    }
    if (!std::strcmp(name, "farn")) {
        if (out)
            *out = DFX_ALGO_FARN;
        return DFX_OK;
    }
    if (!std::strcmp(name, "brox")) {
        if (out)
            *out = DFX_ALGO_BROX;
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
        return fail_create(DFX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (width < 1 || height < 1 || width > 32768 || height > 32768)
        return fail_create(DFX_ERR_INVALID, "invalid frame size");
    if (algo != DFX_ALGO_TVL1 && algo != DFX_ALGO_FARN && algo != DFX_ALGO_BROX)
        return fail_create(DFX_ERR_INVALID, "invalid algorithm id");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail_create(DFX_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail_create(DFX_ERR_INVALID, "device index out of range");

    dfx_context *c = new dfx_context();
    c->device = device;
    c->algo = algo;
    c->W = width;
    c->H = height;
    if (params)
        c->prm = *params;
    else
        default_params(&c->prm);

    auto init = [&]() -> int {
        HIPCHK(c, hipSetDevice(device));
        if (c->prm.blocking_sync) {
            // The runtime's waits spin unless the DEVICE is set to blocking scheduling: hipEventBlockingSync alone left the
            // waiting thread at 100 % CPU (profiles/round4/pmc/blocking_sync_ab.txt).  A process that has already fixed the
            // device's flags (an error here) keeps its own choice.
            (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
            (void)hipGetLastError();
        }
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
        for (auto &e : c->ev_h2d)
            HIPCHK(c, hipEventCreateWithFlags(&e, dfx_event_flags(c, false)));
        for (auto &e : c->ev_compute)
            HIPCHK(c, hipEventCreateWithFlags(&e, dfx_event_flags(c, false)));
        for (auto &e : c->ev_d2h)
            HIPCHK(c, hipEventCreateWithFlags(&e, dfx_event_flags(c, false)));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_t0, dfx_event_flags(c, true)));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_t1, dfx_event_flags(c, true)));
        if (c->prm.blocking_sync)
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_block, dfx_event_flags(c, false)));
        if (algo == DFX_ALGO_TVL1)
            c->engine = dfx_make_tvl1_engine(c);
        else if (algo == DFX_ALGO_FARN)
            c->engine = dfx_make_farneback_engine(c);
        else
            c->engine = dfx_make_brox_engine(c);
        return c->engine->create();
    };
    const int rc = init();
    if (rc != DFX_OK) {
        g_create_error = c->get_err();
        dfx_destroy(c);
        return rc;
    }
    *out = c;
    return DFX_OK;
}

namespace {
// dfx_next_segments applies to the NEXT calc / submit call only, whether that call succeeds or not (include/dfx.h).  The
// list is consumed inside calc_batch_body, which a call rejected by its wrapper's argument checks never reaches: every
// public entry point holds one of these, so a rejected call cannot leave the list armed for an unrelated later one.
struct SegmentsScope {
    dfx_context *c;
    explicit SegmentsScope(dfx_context *ctx) : c(ctx) {}
    ~SegmentsScope() {
        if (c)
            c->next_segments.clear();
    }
    void hand_over() { c = nullptr; } // another entry point takes over (dfx_calc -> dfx_calc_batch)
};
// |step| for the wrappers' early size computations; INT_MIN (whose negation is not an int) saturates, and the body
// rejects it with every other out-of-range step.
inline int abs_step(int step) { return step == INT_MIN ? INT_MAX : std::abs(step); }
} // namespace

int dfx_calc(dfx_handle h, const uint8_t *a, size_t a_pitch, const uint8_t *b, size_t b_pitch, float *flow_uv,
             size_t out_pitch) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    if (!a || !b || !flow_uv)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frame or flow pointer");
    const size_t rb = h->in_row_bytes();
    const int rows = h->in_h();
    if (a_pitch < rb || b_pitch < rb)
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    if (a_pitch != b_pitch) { // dfx_calc_batch takes one pitch: repack both frames densely
        std::vector<uint8_t> ta(rb * rows), tb(rb * rows);
        for (int y = 0; y < rows; ++y) {
            std::memcpy(ta.data() + (size_t)y * rb, a + (size_t)y * a_pitch, rb);
            std::memcpy(tb.data() + (size_t)y * rb, b + (size_t)y * b_pitch, rb);
        }
        const uint8_t *fr[2] = {ta.data(), tb.data()};
        float *fl[1] = {flow_uv};
        seg_scope.hand_over();
        return dfx_calc_batch(h, fr, rb, 2, 1, fl, out_pitch);
    }
    const uint8_t *fr[2] = {a, b};
    float *fl[1] = {flow_uv};
    seg_scope.hand_over();
    return dfx_calc_batch(h, fr, a_pitch, 2, 1, fl, out_pitch);
}

int dfx_calc_batch(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                   float *const *flows_uv, size_t out_pitch) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!frames || !flows_uv))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frames or flows array");
    if (M > 0 && (frame_pitch < h->in_row_bytes() || out_pitch < (size_t)h->W * 8))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    OutSpec out;
    out.flows = flows_uv;
    out.out_pitch = out_pitch;
    return calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, out);
}

int dfx_calc_batch_device(dfx_handle h, const uint8_t *d_frames, size_t pitch, size_t frame_stride, int n_frames,
                          int step, float *d_flows, size_t flow_stride_floats) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!d_frames || !d_flows))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL device frames or flows");
    if (M > 0 && (pitch < h->in_row_bytes() || frame_stride < pitch * (size_t)h->in_h() ||
                  flow_stride_floats < (size_t)h->W * h->H * 2))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch/stride smaller than a frame");
    OutSpec out;
    out.d_flows = d_flows;
    out.d_flow_stride = flow_stride_floats;
    return calc_batch_impl(h, nullptr, 0, d_frames, pitch, frame_stride, n_frames, step, out);
}

int dfx_calc_batch_u8(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                      double lower_bound, double upper_bound, uint8_t *const *img_x, uint8_t *const *img_y,
                      size_t img_pitch) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!frames || !img_x || !img_y))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frames or image plane array");
    if (M > 0 && (frame_pitch < h->in_row_bytes() || img_pitch < (size_t)h->W))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    OutSpec out;
    out.quantized = true;
    out.lo = lower_bound;
    out.hi = upper_bound;
    out.img_x = img_x;
    out.img_y = img_y;
    out.img_pitch = img_pitch;
    return calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, out);
}

int dfx_submit_batch(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                     float *const *flows_uv, size_t out_pitch, uint64_t *ticket) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    if (!ticket)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL ticket");
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!frames || !flows_uv))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frames or flows array");
    if (M > 0 && (frame_pitch < h->in_row_bytes() || out_pitch < (size_t)h->W * 8))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    OutSpec out;
    out.flows = flows_uv;
    out.out_pitch = out_pitch;
    unsigned long long t = 0;
    const int rc = calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, out, &t);
    *ticket = t;
    return rc;
}

int dfx_submit_batch_u8(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                        double lower_bound, double upper_bound, uint8_t *const *img_x, uint8_t *const *img_y,
                        size_t img_pitch, uint64_t *ticket) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    if (!ticket)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL ticket");
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!frames || !img_x || !img_y))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frames or image plane array");
    if (M > 0 && (frame_pitch < h->in_row_bytes() || img_pitch < (size_t)h->W))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    OutSpec out;
    out.quantized = true;
    out.lo = lower_bound;
    out.hi = upper_bound;
    out.img_x = img_x;
    out.img_y = img_y;
    out.img_pitch = img_pitch;
    unsigned long long t = 0;
    const int rc = calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, out, &t);
    *ticket = t;
    return rc;
}

namespace {
int jpeg_entry(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step, double lower_bound,
               double upper_bound, int quality, uint8_t *const *jpg_x, uint8_t *const *jpg_y, size_t jpg_capacity,
               uint32_t *size_x, uint32_t *size_y, uint64_t *ticket) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!frames || !jpg_x || !jpg_y || !size_x || !size_y))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frames, JPEG buffer or size array");
    if (M > 0 && frame_pitch < h->in_row_bytes())
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    if (quality < 1 || quality > 100)
        return dfx_fail(h, DFX_ERR_INVALID, "JPEG quality must be 1..100");
    if (h->W > 65535 || h->H > 65535)
        return dfx_fail(h, DFX_ERR_UNSUPPORTED, "JPEG: frame larger than 65535 pixels");
    OutSpec out;
    out.quantized = true;
    out.jpeg = true;
    out.lo = lower_bound;
    out.hi = upper_bound;
    out.quality = quality;
    out.jpg_x = jpg_x;
    out.jpg_y = jpg_y;
    out.jpg_capacity = jpg_capacity;
    out.size_x = size_x;
    out.size_y = size_y;
    unsigned long long t = 0;
    const int rc = calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, out, ticket ? &t : nullptr);
    if (ticket)
        *ticket = t;
    return rc;
}
} // namespace

// the -st=png scheme (src/common.cpp:18-46, 66-71): planes scaled by the per-flow adaptive bounds + the bounds
static int png_args(dfx_handle h, const void *frames, size_t frame_pitch, int n_frames, int step, const void *img_x,
                    const void *img_y, size_t img_pitch, const void *bounds) {
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!frames || !img_x || !img_y || !bounds))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frames, image plane array or bounds array");
    if (M > 0 && (frame_pitch < h->in_row_bytes() || img_pitch < (size_t)h->W))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    return DFX_OK;
}
int dfx_calc_batch_png(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                       uint8_t *const *img_x, uint8_t *const *img_y, size_t img_pitch, double *bounds_xy) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    const int arc = png_args(h, frames, frame_pitch, n_frames, step, img_x, img_y, img_pitch, bounds_xy);
    if (arc != DFX_OK)
        return arc;
    OutSpec out;
    out.quantized = out.png = true;
    out.img_x = img_x, out.img_y = img_y, out.img_pitch = img_pitch, out.bounds = bounds_xy;
    return calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, out);
}
int dfx_submit_batch_png(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                         uint8_t *const *img_x, uint8_t *const *img_y, size_t img_pitch, double *bounds_xy,
                         uint64_t *ticket) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    if (!ticket)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL ticket");
    const int arc = png_args(h, frames, frame_pitch, n_frames, step, img_x, img_y, img_pitch, bounds_xy);
    if (arc != DFX_OK)
        return arc;
    OutSpec out;
    out.quantized = out.png = true;
    out.img_x = img_x, out.img_y = img_y, out.img_pitch = img_pitch, out.bounds = bounds_xy;
    unsigned long long t = 0;
    const int rc = calc_batch_impl(h, frames, frame_pitch, nullptr, 0, 0, n_frames, step, out, &t);
    *ticket = t;
    return rc;
}
int dfx_calc_batch_png_device(dfx_handle h, const uint8_t *d_frames, size_t pitch, size_t frame_stride, int n_frames,
                              int step, uint8_t *d_img_x, uint8_t *d_img_y, size_t img_pitch, size_t img_stride,
                              double *d_bounds_xy) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!d_frames || !d_img_x || !d_img_y || !d_bounds_xy))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL device frames, image planes or bounds");
    if (M > 0 && (pitch < h->in_row_bytes() || frame_stride < pitch * (size_t)h->in_h() || img_pitch < (size_t)h->W ||
                  img_stride < img_pitch * (size_t)h->H))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch/stride smaller than a frame");
    OutSpec out;
    out.quantized = out.png = true;
    out.d_img_x = d_img_x, out.d_img_y = d_img_y, out.img_pitch = img_pitch, out.d_img_stride = img_stride;
    out.d_bounds = d_bounds_xy;
    return calc_batch_impl(h, nullptr, 0, d_frames, pitch, frame_stride, n_frames, step, out);
}

int dfx_calc_batch_jpeg(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                        double lower_bound, double upper_bound, int quality, uint8_t *const *jpg_x,
                        uint8_t *const *jpg_y, size_t jpg_capacity, uint32_t *size_x, uint32_t *size_y) {
    return jpeg_entry(h, frames, frame_pitch, n_frames, step, lower_bound, upper_bound, quality, jpg_x, jpg_y,
                      jpg_capacity, size_x, size_y, nullptr);
}

int dfx_submit_batch_jpeg(dfx_handle h, const uint8_t *const *frames, size_t frame_pitch, int n_frames, int step,
                          double lower_bound, double upper_bound, int quality, uint8_t *const *jpg_x,
                          uint8_t *const *jpg_y, size_t jpg_capacity, uint32_t *size_x, uint32_t *size_y,
                          uint64_t *ticket) {
    if (h && !ticket) {
        h->next_segments.clear();
        return dfx_fail(h, DFX_ERR_INVALID, "NULL ticket");
    }
    return jpeg_entry(h, frames, frame_pitch, n_frames, step, lower_bound, upper_bound, quality, jpg_x, jpg_y,
                      jpg_capacity, size_x, size_y, ticket);
}

int dfx_encode_jpeg(dfx_handle h, const uint8_t *const *planes, size_t pitch, int n, int quality, uint8_t *const *jpg,
                    size_t jpg_capacity, uint32_t *sizes) {
    if (!h)
        return DFX_ERR_INVALID;
    (void)dfx_finish_tails(h, 0, -1);
    if (n < 0)
        return dfx_fail(h, DFX_ERR_INVALID, "n must be >= 0");
    if (n == 0)
        return DFX_OK;
    if (!planes || !jpg || !sizes)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL plane, JPEG buffer or size array");
    if (pitch < (size_t)h->W)
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    if (quality < 1 || quality > 100)
        return dfx_fail(h, DFX_ERR_INVALID, "JPEG quality must be 1..100");
    HIPCHK(h, hipSetDevice(h->device));
    const int B = h->engine->batch();
    int rc = ensure_img_staging(h, B);
    if (rc == DFX_OK)
        rc = ensure_jpeg(h, B, quality);
    if (rc != DFX_OK)
        return rc;
    const size_t plane = (size_t)h->W * h->H;
    const int chunk_max = 2 * h->img_slots; // a staging set holds that many planes back to back
    for (int i0 = 0; i0 < n; i0 += chunk_max) {
        const int nc = std::min(chunk_max, n - i0);
        for (int j = 0; j < nc; ++j)
            HIPCHK(h, hipMemcpy2DAsync(h->d_img[0] + (size_t)j * plane, (size_t)h->W, planes[i0 + j], pitch, (size_t)h->W,
                                       (size_t)h->H, hipMemcpyHostToDevice, h->stream));
        JpegCtx jc;
        jc.planes = h->d_img[0];
        jc.plane_stride = (long long)plane;
        jc.pitch = h->W, jc.w = h->W, jc.h = h->H, jc.bw = (h->W + 7) / 8, jc.bh = (h->H + 7) / 8;
        jc.n_planes = nc, jc.n_x = nc, jc.y_first = 0;
        jc.tab = h->jpeg.d_tab, jc.dc = h->jpeg.d_dc, jc.bits = h->jpeg.d_bits;
        jc.plane_bits = h->jpeg.d_plane_bits, jc.plane_base = h->jpeg.d_plane_base;
        jc.stream = h->jpeg.d_stream[0], jc.capacity_bytes = h->jpeg.capacity;
        jc.info = h->jpeg.d_info[0], jc.hdr = h->jpeg.d_hdr;
        jpeg_launch_encode(h->stream, jc);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const unsigned long long *hi = h->jpeg.h_info[0];
        if (hi[1]) { // more than the 4 bits per pixel the shared buffer was sized for: grow it to what the scan pass measured
            const int grc = grow_jpeg_streams(h, hi[0]);
            if (grc != DFX_OK)
                return grc;
            jc.stream = h->jpeg.d_stream[0], jc.capacity_bytes = h->jpeg.capacity;
            jpeg_launch_encode(h->stream, jc);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        if (hi[1])
            return dfx_fail(h, DFX_ERR_UNSUPPORTED,
                            "JPEG: the planes do not fit the stream buffer (encode them on the host)");
        {
            const int lrc = ensure_jpeg_landing(h, 0, (size_t)hi[0]); // synchronous entry point: no tail is in flight
            if (lrc != DFX_OK)
                return lrc;
        }
        HIPCHK(h, hipMemcpyAsync(h->jpeg.h_stream[0], h->jpeg.d_stream[0], (size_t)hi[0], hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int j = 0; j < nc; ++j) {
            const size_t sz = jpeg_assemble(h->jpeg.header, h->jpeg.h_stream[0] + hi[2 + 2 * j + 1], hi[2 + 2 * j], jpg[i0 + j],
                                            jpg_capacity);
            if (sz == 0)
                return dfx_fail(h, DFX_ERR_UNSUPPORTED, "JPEG: jpg_capacity is too small for an encoded plane");
            sizes[i0 + j] = (uint32_t)sz;
        }
    }
    return DFX_OK;
}

size_t dfx_jpeg_capacity(dfx_handle h) {
    if (!h)
        return 0;
    // header (623 bytes) + the entropy-coded segment.  A plane whose segment exceeds its pixel count (8 bits per pixel
    // BEFORE stuffing) is far outside what flow images produce; such a FlowBuffer fails with DFX_ERR_UNSUPPORTED and the
    // caller encodes its 8-bit planes (dfx_calc_batch_u8) itself.
    return (size_t)h->W * h->H + 4096;
}

int dfx_next_segments(dfx_handle h, const int *seg_frames, int n_segments) {
    if (!h)
        return DFX_ERR_INVALID;
    h->next_segments.clear();
    if (n_segments < 0 || (n_segments > 0 && !seg_frames))
        return dfx_fail(h, DFX_ERR_INVALID, "dfx_next_segments: NULL clip lengths");
    for (int i = 0; i < n_segments; ++i) {
        if (seg_frames[i] < 0) {
            h->next_segments.clear();
            return dfx_fail(h, DFX_ERR_INVALID, "dfx_next_segments: negative clip length");
        }
        h->next_segments.push_back(seg_frames[i]);
    }
    return DFX_OK;
}

int dfx_wait(dfx_handle h, uint64_t ticket) {
    if (!h)
        return DFX_ERR_INVALID;
    return dfx_finish_tails(h, ticket, -1, /*report=*/true);
}

int dfx_calc_batch_u8_device(dfx_handle h, const uint8_t *d_frames, size_t pitch, size_t frame_stride, int n_frames,
                             int step, double lower_bound, double upper_bound, uint8_t *d_img_x, uint8_t *d_img_y,
                             size_t img_pitch, size_t img_stride) {
    if (!h)
        return DFX_ERR_INVALID;
    SegmentsScope seg_scope(h);
    const int M = std::max(n_frames - abs_step(step), 0);
    if (M > 0 && (!d_frames || !d_img_x || !d_img_y))
        return dfx_fail(h, DFX_ERR_INVALID, "NULL device frames or image planes");
    if (M > 0 && (pitch < h->in_row_bytes() || frame_stride < pitch * (size_t)h->in_h() || img_pitch < (size_t)h->W ||
                  img_stride < img_pitch * (size_t)h->H))
        return dfx_fail(h, DFX_ERR_INVALID, "pitch/stride smaller than a frame");
    OutSpec out;
    out.quantized = true;
    out.lo = lower_bound;
    out.hi = upper_bound;
    out.d_img_x = d_img_x;
    out.d_img_y = d_img_y;
    out.img_pitch = img_pitch;
    out.d_img_stride = img_stride;
    return calc_batch_impl(h, nullptr, 0, d_frames, pitch, frame_stride, n_frames, step, out);
}

int dfx_flow_to_u8_device(dfx_handle h, const float *d_flows, size_t flow_stride_floats, int n, double lower_bound,
                          double upper_bound, uint8_t *d_img_x, uint8_t *d_img_y, size_t img_pitch,
                          size_t img_stride) {
    if (!h)
        return DFX_ERR_INVALID;
    (void)dfx_finish_tails(h, 0, -1);
    if (n < 0)
        return dfx_fail(h, DFX_ERR_INVALID, "n must be >= 0");
    if (n == 0)
        return DFX_OK;
    if (!d_flows || !d_img_x || !d_img_y)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL device flows or image planes");
    if (flow_stride_floats < (size_t)h->W * h->H * 2 || img_pitch < (size_t)h->W ||
        img_stride < img_pitch * (size_t)h->H)
        return dfx_fail(h, DFX_ERR_INVALID, "pitch/stride smaller than a frame");
    HIPCHK(h, hipSetDevice(h->device));
    quant_launch_flow_to_u8(h->stream, d_flows, (long long)flow_stride_floats, n, h->W, h->H, lower_bound,
                            upper_bound, d_img_x, d_img_y, (long long)img_pitch, (long long)img_stride);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DFX_OK;
}

int dfx_flow_to_png_device(dfx_handle h, const float *d_flows, size_t flow_stride_floats, int n, uint8_t *d_img_x,
                           uint8_t *d_img_y, size_t img_pitch, size_t img_stride, double *d_bounds_xy) {
    if (!h)
        return DFX_ERR_INVALID;
    (void)dfx_finish_tails(h, 0, -1);
    if (n < 0)
        return dfx_fail(h, DFX_ERR_INVALID, "n must be >= 0");
    if (n == 0)
        return DFX_OK;
    if (!d_flows || !d_img_x || !d_img_y || !d_bounds_xy)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL device flows, image planes or bounds");
    if (flow_stride_floats < (size_t)h->W * h->H * 2 || img_pitch < (size_t)h->W ||
        img_stride < img_pitch * (size_t)h->H)
        return dfx_fail(h, DFX_ERR_INVALID, "pitch/stride smaller than a frame");
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = ensure_png(h, n);
    if (rc != DFX_OK)
        return rc;
    quant_launch_flow_to_png_planes(h->stream, d_flows, (long long)flow_stride_floats, n, h->W, h->H, h->d_png_scratch,
                                    d_bounds_xy, d_img_x, d_img_y, (long long)img_pitch, (long long)img_stride);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DFX_OK;
}

int dfx_set_source_format(dfx_handle h, int src_width, int src_height, int channels) {
    if (!h)
        return DFX_ERR_INVALID;
    if (src_width == 0 && src_height == 0) { // back to the default: W x H gray frames
        h->src_w = h->src_h = 0;
        h->src_ch = 1;
        return DFX_OK;
    }
    if (src_width < 1 || src_height < 1 || src_width > 32768 || src_height > 32768)
        return dfx_fail(h, DFX_ERR_INVALID, "invalid source frame size");
    if (channels != 1 && channels != 3)
        return dfx_fail(h, DFX_ERR_INVALID, "channels must be 1 (gray) or 3 (BGR)");
    if (src_width == h->W && src_height == h->H && channels == 1) {
        h->src_w = h->src_h = 0;
        h->src_ch = 1;
        return DFX_OK;
    }
    h->src_w = src_width;
    h->src_h = src_height;
    h->src_ch = channels;
    return DFX_OK;
}

int dfx_prepare_frames_device(dfx_handle h, const uint8_t *d_src, size_t src_pitch, size_t src_frame_stride,
                              int src_width, int src_height, int channels, int n, uint8_t *d_gray, size_t gray_pitch,
                              size_t gray_frame_stride) {
    if (!h)
        return DFX_ERR_INVALID;
    (void)dfx_finish_tails(h, 0, -1);
    if (n < 0)
        return dfx_fail(h, DFX_ERR_INVALID, "n must be >= 0");
    if (n == 0)
        return DFX_OK;
    if (!d_src || !d_gray)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL device frames");
    if (src_width < 1 || src_height < 1 || (channels != 1 && channels != 3))
        return dfx_fail(h, DFX_ERR_INVALID, "invalid source format");
    if (src_pitch < (size_t)src_width * channels || src_frame_stride < src_pitch * (size_t)src_height ||
        gray_pitch < (size_t)h->W || gray_frame_stride < gray_pitch * (size_t)h->H)
        return dfx_fail(h, DFX_ERR_INVALID, "pitch/stride smaller than a frame");
    HIPCHK(h, hipSetDevice(h->device));
    prepare_launch(h->stream, d_src, (long long)src_pitch, (long long)src_frame_stride, src_width, src_height, channels,
                   n, d_gray, (long long)gray_pitch, (long long)gray_frame_stride, h->W, h->H);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DFX_OK;
}

int dfx_prepare_frames(dfx_handle h, const uint8_t *const *src, size_t src_pitch, int src_width, int src_height,
                       int channels, int n, uint8_t *const *gray, size_t gray_pitch) {
    if (!h)
        return DFX_ERR_INVALID;
    (void)dfx_finish_tails(h, 0, -1);
    if (n < 0)
        return dfx_fail(h, DFX_ERR_INVALID, "n must be >= 0");
    if (n == 0)
        return DFX_OK;
    if (!src || !gray)
        return dfx_fail(h, DFX_ERR_INVALID, "NULL frame arrays");
    if (src_width < 1 || src_height < 1 || (channels != 1 && channels != 3))
        return dfx_fail(h, DFX_ERR_INVALID, "invalid source format");
    const size_t rb = (size_t)src_width * channels, fb = rb * src_height, plane = (size_t)h->W * h->H;
    if (src_pitch < rb || gray_pitch < (size_t)h->W)
        return dfx_fail(h, DFX_ERR_INVALID, "pitch smaller than a row");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned char *d_in = nullptr, *d_out = nullptr;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)256 << 20) / std::max(fb, plane)));
    auto run = [&]() -> int {
        HIPCHK(h, hipMalloc(&d_in, (size_t)chunk * fb));
        HIPCHK(h, hipMalloc(&d_out, (size_t)chunk * plane));
        for (int i0 = 0; i0 < n; i0 += chunk) {
            const int m = std::min(chunk, n - i0);
            for (int j = 0; j < m; ++j)
                HIPCHK(h, hipMemcpy2DAsync(d_in + (size_t)j * fb, rb, src[i0 + j], src_pitch, rb, src_height,
                                           hipMemcpyHostToDevice, h->stream));
            prepare_launch(h->stream, d_in, (long long)rb, (long long)fb, src_width, src_height, channels, m, d_out,
                           h->W, (long long)plane, h->W, h->H);
            HIPCHK(h, hipGetLastError());
            for (int j = 0; j < m; ++j)
                HIPCHK(h, hipMemcpy2DAsync(gray[i0 + j], gray_pitch, d_out + (size_t)j * plane, h->W, h->W, h->H,
                                           hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        return DFX_OK;
    };
    const int rc = run();
    dfx_free_dev(d_in);
    dfx_free_dev(d_out);
    return rc;
}

int dfx_get_stats(dfx_handle h, dfx_stats *out) {
    if (!h || !out)
        return DFX_ERR_INVALID;
    *out = h->stats;
    out->batch = h->engine ? h->engine->batch() : 0;
    return DFX_OK;
}

void dfx_reset_stats(dfx_handle h) {
    if (h)
        std::memset(&h->stats, 0, sizeof h->stats);
}

const char *dfx_last_error(dfx_handle h) {
    if (!h)
        return g_create_error.c_str();
    thread_local std::string text; // a copy per calling thread: the collector and the owner may both ask
    text = h->get_err();
    return text.c_str();
}

void dfx_destroy(dfx_handle h) {
    if (!h)
        return;
    (void)hipSetDevice(h->device);
    (void)dfx_finish_tails(h, 0, -1);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    if (h->d2h_stream)
        (void)hipStreamSynchronize(h->d2h_stream);
    delete h->engine;
    h->engine = nullptr;
    if (h->copy_stream)
        (void)hipStreamSynchronize(h->copy_stream);
    for (auto &p : h->d_u8)
        dfx_free_dev(p);
    for (auto &p : h->d_flow_out)
        dfx_free_dev(p);
    for (auto &p : h->d_img)
        dfx_free_dev(p);
    for (auto &p : h->d_src)
        dfx_free_dev(p);
    for (auto &p : h->h_in)
        dfx_free_host(p);
    for (auto &p : h->h_out)
        dfx_free_host(p);
    dfx_free_dev(h->d_png_scratch);
    for (auto &p : h->h_png_bounds)
        dfx_free_host(p);
    free_jpeg(h);
    for (auto &e : h->ev_h2d)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : h->ev_compute)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : h->ev_d2h)
        if (e)
            (void)hipEventDestroy(e);
    if (h->copy_stream)
        (void)hipStreamDestroy(h->copy_stream);
    if (h->d2h_stream)
        (void)hipStreamDestroy(h->d2h_stream);
    if (h->ev_block)
        (void)hipEventDestroy(h->ev_block);
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
    // portable: the host shell's loader threads allocate frames that any device's handle may copy from
    return hipHostMalloc(ptr, bytes, hipHostMallocPortable) == hipSuccess ? DFX_OK : DFX_ERR_HIP;
}

int dfx_host_free(void *ptr) { return hipHostFree(ptr) == hipSuccess ? DFX_OK : DFX_ERR_HIP; }

} // extern "C"
