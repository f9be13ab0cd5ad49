// tests/host_harness.cpp — TEST INFRASTRUCTURE: C wrappers around the host shell's codecs/quantisation and
// the DenseFlow operator (calc_optflows_imp) so pytest can drive them through ctypes.
#include <chrono>
#include <cstring>

#include "../include/dense_flow.h"
#include "../include/dfx_jpeg_tables.h"
#include "../include/utils.h"

extern "C" {

// resizeLinear (the host-side cv::resize stand-in) on a dense gray frame
void hh_resize(const uchar *src, int sw, int sh, uchar *dst, int dw, int dh) {
    Mat a(Size(sw, sh), CV_8UC1), b;
    memcpy(a.data(), src, (size_t)sw * sh);
    resizeLinear(a, b, Size(dw, dh));
    memcpy(dst, b.data(), (size_t)dw * dh);
}

void hh_jpeg_force_portable(int on) { imencodeJpegForcePortable(on != 0); }

int hh_encode_jpeg(const uchar *gray, int w, int h, int quality, uchar *out, int out_cap) {
    Mat m(Size(w, h), CV_8UC1);
    memcpy(m.data(), gray, (size_t)w * h);
    vector<uchar> buf;
    if (!imencodeJpeg(m, buf, quality) || (int)buf.size() > out_cap)
        return -1;
    memcpy(out, buf.data(), buf.size());
    return (int)buf.size();
}

// dfx_jpeg_quantise (the reciprocal-multiply form both encoders use) against libjpeg's rule written with a plain
// integer division, for every divisor 8 q (q = 1..255) and every transform output in +-limit: the number of mismatches
long long hh_jpeg_quantise_mismatches(int limit) {
    long long bad = 0;
    for (unsigned q = 1; q <= 255; ++q) {
        const unsigned d = 8 * q, m = dfx_jpeg_divide_magic(d);
        for (int c = -limit; c <= limit; ++c) {
            int t = c < 0 ? -c : c;
            t = (t + (int)(d >> 1)) / (int)d;
            bad += dfx_jpeg_quantise(c, d, m) != (c < 0 ? -t : t);
        }
    }
    return bad;
}

// imencodePng of an 8-bit gray (ch = 1) or BGR (ch = 3) image
int hh_encode_png(const uchar *px, int w, int h, int ch, uchar *out, int out_cap) {
    Mat m(Size(w, h), ch == 1 ? CV_8UC1 : CV_8UC3);
    memcpy(m.data(), px, (size_t)w * h * ch);
    vector<uchar> buf;
    if (!imencodePng(m, buf) || (int)buf.size() > out_cap)
        return -1;
    memcpy(out, buf.data(), buf.size());
    return (int)buf.size();
}

int hh_encode_flow_png(const float *fx, const float *fy, int w, int h, uchar *out, int out_cap) {
    Mat a(Size(w, h), CV_32FC1), b(Size(w, h), CV_32FC1);
    memcpy(a.data(), fx, sizeof(float) * w * h);
    memcpy(b.data(), fy, sizeof(float) * w * h);
    vector<uchar> buf;
    encodeFlowMapPng(a, b, buf);
    if ((int)buf.size() > out_cap)
        return -1;
    memcpy(out, buf.data(), buf.size());
    return (int)buf.size();
}

void hh_quantise(const float *fx, const float *fy, int w, int h, int bound, uchar *ox, uchar *oy) {
    Mat a(Size(w, h), CV_32FC1), b(Size(w, h), CV_32FC1), ia(Size(w, h), CV_8UC1), ib(Size(w, h), CV_8UC1);
    memcpy(a.data(), fx, sizeof(float) * w * h);
    memcpy(b.data(), fy, sizeof(float) * w * h);
    convertFlowToImage(a, b, ia, ib, -bound, bound);
    memcpy(ox, ia.data(), (size_t)w * h);
    memcpy(oy, ib.data(), (size_t)w * h);
}

// DenseFlow::calc_optflows_imp on n in-memory frames; flows: (n-|step|) x h x w x 2 floats. Returns #flows or <0.
int hh_calc_optflows_imp(const uchar *frames, int n, int w, int h, const char *algorithm, int step, float *flows,
                         char *err, int err_cap) {
    try {
        vector<path> none;
        DenseFlow d(none, none, "tvl1", step, 20, 0, 0, 0, false, false, "jpg");
        vector<Mat> fr(n);
        for (int i = 0; i < n; ++i) {
            fr[i].create(Size(w, h), CV_8UC1);
            memcpy(fr[i].data(), frames + (size_t)i * w * h, (size_t)w * h);
        }
        vector<Mat> out = DenseFlowTestAccess::run_calc_optflows_imp(d, fr, algorithm, step);
        for (size_t i = 0; i < out.size(); ++i)
            memcpy(flows + i * (size_t)w * h * 2, out[i].data(), sizeof(float) * w * h * 2);
        return (int)out.size();
    } catch (const std::exception &e) {
        snprintf(err, err_cap, "%s", e.what());
        return -1;
    }
}

// The same with the bounding done on the device (what launch() does for save_type "jpg"):
// planes: (n-|step|) x 2 x h x w bytes (x plane, then y plane, per flow).  Returns #flows or <0.
int hh_calc_optflows_imp_bounded(const uchar *frames, int n, int w, int h, const char *algorithm, int step, int bound,
                                 uchar *planes, char *err, int err_cap) {
    try {
        vector<path> none;
        DenseFlow d(none, none, "tvl1", step, bound, 0, 0, 0, false, false, "jpg");
        vector<Mat> fr(n);
        for (int i = 0; i < n; ++i) {
            fr[i].create(Size(w, h), CV_8UC1);
            memcpy(fr[i].data(), frames + (size_t)i * w * h, (size_t)w * h);
        }
        vector<Mat> out = DenseFlowTestAccess::run_calc_optflows_imp(d, fr, algorithm, step, true);
        for (size_t i = 0; i < out.size(); ++i)
            memcpy(planes + i * (size_t)w * h, out[i].data(), (size_t)w * h);
        return (int)out.size() / 2;
    } catch (const std::exception &e) {
        snprintf(err, err_cap, "%s", e.what());
        return -1;
    }
}

// N threads open the same .y4m clip `rounds` times each (the loader threads of a multi-device run do that
// concurrently); returns the number of opens whose parsed geometry / frame count were wrong.
int hh_parallel_open_y4m(const char *file, int threads, int rounds, int w, int h, int frames) {
    std::atomic<int> bad(0);
    parallelFor(threads, threads, [&](int) {
        for (int r = 0; r < rounds; ++r) {
            VideoCapture cap;
            Mat first;
            if (!cap.open(file) || cap.width() != w || cap.height() != h || cap.frameCount() != frames ||
                !cap.read(first) || first.cols != w || first.rows != h)
                bad += 1;
        }
    });
    return bad.load();
}

// N threads encode the same gray image to JPEG; returns 1 when every stream is byte-identical to the serial one.
int hh_parallel_jpeg(const uchar *gray, int w, int h, int threads) {
    Mat m(Size(w, h), CV_8UC1);
    memcpy(m.data(), gray, (size_t)w * h);
    vector<uchar> ref;
    imencodeJpeg(m, ref);
    vector<vector<uchar>> outs(threads * 4);
    parallelFor((int)outs.size(), threads, [&](int i) { imencodeJpeg(m, outs[i]); });
    for (auto &o : outs)
        if (o != ref)
            return 0;
    return 1;
}

// FlowBufferQueue: a producer pushes n buffers (base_start = i, the last one final) through a queue of depth `depth`
// while this thread pops until the final flag; returns the sum of the popped base_start values, or -1 on disorder.
long hh_queue_roundtrip(int n, int depth) {
    FlowBufferQueue q((size_t)depth);
    thread producer([&] {
        for (int i = 0; i < n; ++i)
            q.push(FlowBuffer({}, path(), i, false), i == n - 1);
    });
    long sum = 0;
    int expect = 0;
    bool ok = true;
    while (true) {
        bool fin = false;
        FlowBuffer b = q.pop(&fin);
        ok = ok && b.base_start == expect++;
        sum += b.base_start;
        if (fin)
            break;
    }
    producer.join();
    return ok && expect == n ? sum : -1;
}

// close(): a consumer blocked on an empty queue wakes up with an empty final buffer, a producer blocked on a full
// queue returns.  1 = both happened.
int hh_queue_close_unblocks() {
    FlowBufferQueue empty_q(2), full_q(1);
    full_q.push(FlowBuffer({}, path(), 0, false), false);
    std::atomic<int> woke(0);
    thread consumer([&] {
        bool fin = false;
        FlowBuffer b = empty_q.pop(&fin);
        if (fin && b.item_data.empty())
            woke += 1;
    });
    thread producer([&] {
        full_q.push(FlowBuffer({}, path(), 1, false), false); // blocks: depth 1 and nobody pops
        woke += 1;
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    empty_q.close();
    full_q.close();
    consumer.join();
    producer.join();
    return woke.load() == 2;
}

// try_pop and the byte budget: a depth-1 queue with a budget of `budget` bytes takes small buffers (each `frame_bytes`) until
// the budget (or hard_max) is reached, try_pop drains it in order and then reports "nothing queued".  Returns the number of
// buffers a producer could push without blocking, or -1 on disorder.
int hh_queue_budget(int frame_bytes, int budget, int hard_max) {
    FlowBufferQueue q(1);
    q.set_byte_budget((size_t)budget, (size_t)hard_max);
    std::atomic<int> pushed(0);
    thread producer([&] {
        for (int i = 0; i < 1000; ++i) {
            vector<Mat> frames(1);
            frames[0].create(Size(frame_bytes, 1), CV_8UC1);
            q.push(FlowBuffer(std::move(frames), path(), i, false), false);
            pushed += 1;
        }
    });
    int last = -1;
    for (int spins = 0; spins < 200; ++spins) { // wait until the producer is blocked: the count stops growing
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        const int now = pushed.load();
        if (now == last)
            break;
        last = now;
    }
    const int accepted = pushed.load();
    bool ok = true;
    int expect = 0;
    FlowBuffer b({}, path(), 0, false);
    bool fin = false;
    for (int i = 0; i < accepted; ++i)
        ok = ok && q.try_pop(b, &fin) && b.base_start == expect++ && !fin;
    q.close(); // the producer's blocked push returns, later pushes are dropped
    producer.join();
    while (q.try_pop(b, &fin)) { // what slipped in between the drain and close()
    }
    ok = ok && !q.try_pop(b, &fin);
    return ok ? accepted : -1;
}

// parallelFor: sum of i over [0, n) computed on `threads` workers; throws_at >= 0 makes that index throw.
long hh_parallel_sum(int n, int threads, int throws_at) {
    std::atomic<long> sum(0);
    try {
        parallelFor(n, threads, [&](int i) {
            if (i == throws_at)
                throw std::runtime_error("boom");
            sum += i;
        });
    } catch (const std::exception &) {
        return -1;
    }
    return sum.load();
}
}

extern "C" {
// DenseFlow::shard_range (the shell's Level-2 split) for the equivalence test with denseflow_amd/shard.py
void hh_shard_range(int n_frames, int step, int rank, int world, int *begin, int *end) {
    DenseFlow::shard_range(n_frames, step, rank, world, *begin, *end);
}
}
