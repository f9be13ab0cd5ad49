// src/denseflow_gpu.cpp — the host shell: denseflow's three-stage pipeline (load -> flow -> encode/save)
// with the flow stage running on MI355X through the C ABI (include/dfx.h).
//
// Mirrors /root/reference/src/denseflow_gpu.cpp: check_param :9-42 (same messages), get_new_size :44-80,
// extract_frames_only :82-144, load_frames_batch/video :146-217, load_frames :219-280,
// calc_optflows_imp :282-370 (the hot path; here one dfx_calc_batch per FlowBuffer),
// calc_optflows :372-394, encode_save :396-477, calcDenseFlowVideoGPU :479-497 (same summary line).
// Re-designed rather than transcribed: bounded queues are a small class, the "ready_to_exit" flags
// travel with the buffers (no unsynchronised bools), media I/O is decoder-free (image_io.h).
#include "dense_flow.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <cmath>

#include "utils.h"

#include <sys/wait.h>
#include <unistd.h>

// DF_TRACE=1: stage-by-stage trace on stderr (debugging aid; the reference has none)
static const bool g_trace = std::getenv("DF_TRACE") != nullptr;
static double trace_ms() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define TRACE(...)                                                                                              \
    do {                                                                                                        \
        if (g_trace) {                                                                                          \
            fprintf(stderr, "[denseflow %9.3f ms] ", trace_ms());                                               \
            fprintf(stderr, __VA_ARGS__);                                                                       \
            fputc('\n', stderr);                                                                                \
            fflush(stderr);                                                                                     \
        }                                                                                                       \
    } while (0)

// adds the scope's wall time to a stage counter (DenseFlow::StageTimes)
struct StageTick {
    long long &acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit StageTick(long long &a) : acc(a) {}
    ~StageTick() { acc += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// ------------------------------------------------------------------------------------------------ queue

static size_t flowbuffer_bytes(const FlowBuffer &b) {
    size_t n = 0;
    for (const Mat &m : b.item_data)
        n += (size_t)m.rows * m.step;
    if (b.encoded)
        for (uint32_t s : b.encoded->size_x)
            n += 2 * (size_t)s;
    return n;
}

void FlowBufferQueue::push(FlowBuffer b, bool is_final) {
    unique_lock<mutex> lock(mtx_);
    not_full_.wait(lock, [&] {
        return closed_ || q_.size() < maxsize_ || (bytes_ < byte_budget_ && q_.size() < hard_max_);
    });
    if (closed_)
        return;
    const size_t nb = flowbuffer_bytes(b);
    bytes_ += nb;
    q_bytes_.push(nb);
    q_.emplace(std::move(b), is_final);
    not_empty_.notify_all();
}

bool FlowBufferQueue::try_pop(FlowBuffer &out, bool *was_final) {
    unique_lock<mutex> lock(mtx_);
    if (q_.empty())
        return false;
    std::pair<FlowBuffer, bool> item = std::move(q_.front());
    q_.pop();
    bytes_ -= q_bytes_.front();
    q_bytes_.pop();
    not_full_.notify_all();
    *was_final = item.second;
    out = std::move(item.first);
    return true;
}

void FlowBufferQueue::close() {
    unique_lock<mutex> lock(mtx_);
    closed_ = true;
    not_full_.notify_all();
    not_empty_.notify_all();
}

size_t FlowBufferQueue::size() {
    unique_lock<mutex> lock(mtx_);
    return q_.size();
}

FlowBuffer FlowBufferQueue::pop(bool *was_final) {
    unique_lock<mutex> lock(mtx_);
    not_empty_.wait(lock, [&] { return closed_ || !q_.empty(); });
    if (q_.empty()) { // closed
        *was_final = true;
        return FlowBuffer({}, path(), 0, false);
    }
    std::pair<FlowBuffer, bool> item = std::move(q_.front());
    q_.pop();
    bytes_ -= q_bytes_.front();
    q_bytes_.pop();
    not_full_.notify_all();
    *was_final = item.second;
    return std::move(item.first);
}

// ------------------------------------------------------------------------------------------------ ctor / params

DenseFlow::DenseFlow(vector<path> video_paths, vector<path> output_dirs, string algorithm, int step, int bound,
                     int new_width, int new_height, int new_short, bool has_class, bool is_record, string save_type,
                     int device)
    : video_paths(std::move(video_paths)), output_dirs(std::move(output_dirs)), algorithm(std::move(algorithm)),
      save_type(std::move(save_type)), step(step), bound(bound), new_width(new_width), new_height(new_height),
      new_short(new_short), has_class(has_class), is_record(is_record), device(device), batch_maxsize(512),
      frames_gray_queue(3), flows_queue(3), total_frames(0), total_flows(0), dfx_(nullptr) {
    device_bounding = this->save_type == "jpg" && !std::getenv("DF_HOST_BOUND");
    device_jpeg = device_bounding && !std::getenv("DF_HOST_JPEG");
    device_png = this->save_type == "png" && !std::getenv("DF_HOST_PNG");
    device_resize = !std::getenv("DF_HOST_RESIZE");
    join_short_ = !std::getenv("DF_NO_JOIN");
    // short clips: let a few of them queue up so that the flow stage can join them (large FlowBuffers stay at 3 per queue)
    frames_gray_queue.set_byte_budget(256u << 20, 48);
    flows_queue.set_byte_budget(256u << 20, 48);
    const char *et = std::getenv("DF_ENCODE_THREADS");
    int hw = (int)std::thread::hardware_concurrency();
    // a container CPU quota (cgroup v2 cpu.max) is the real core count: the MI355X boxes of this pool show 256
    // logical CPUs and allow 16; 32 busy encoder threads there only get throttled
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long q = 0, per = 0;
        if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0)
            hw = std::max(1, std::min(hw, (int)(q / per)));
        fclose(f);
    }
    encode_threads = et ? std::max(1, std::atoi(et)) : std::max(1, std::min(hw > 0 ? hw : 1, 32));
    if (!check_param())
        throw std::runtime_error("check init param error.");
}

DenseFlow::~DenseFlow() {
    if (dfx_)
        dfx_destroy(dfx_);
}

bool DenseFlow::check_param() {
    for (size_t i = 0; i < video_paths.size(); i++) {
        if (!exists(video_paths[i])) {
            cout << video_paths[i] << " does not exist!" << endl;
            return false;
        }
        if (!is_directory(output_dirs[i])) {
            cout << output_dirs[i] << " is not a valid dir!" << endl;
            return false;
        }
    }
    if (algorithm != "nv" && algorithm != "tvl1" && algorithm != "farn" && algorithm != "brox") {
        cout << algorithm << " not supported!" << endl;
        return false;
    }
    if (bound <= 0) {
        cout << "bound should > 0!" << endl;
        return false;
    }
    if (step < -(1 << 24) || step > (1 << 24)) { // keeps |step| and frame-index arithmetic in range; not in the reference
        cout << "step out of range!" << endl;
        return false;
    }
    if (new_height < 0 || new_width < 0 || new_short < 0) {
        cout << "height and width cannot < 0!" << endl;
        return false;
    }
    if (new_height > 32768 || new_width > 32768 || new_short > 32768) { // the engine's limit (dfx_create); not in the reference
        cout << "height and width cannot > 32768!" << endl;
        return false;
    }
    if (new_short > 0 && new_height + new_width != 0) {
        cout << "do not set height and width when set short!" << endl;
        return false;
    }
    if (save_type != "jpg" && save_type != "png" && save_type != "h5") {
        cout << "only jpg/png/h5 are supported (no " << save_type << ") for output" << endl;
        return false;
    }
    return true;
}

bool DenseFlow::get_new_size(const VideoCapture &video_stream, const vector<path> &frames_path, bool use_frames,
                             Size &new_size, int &frames_num) {
    int width, height;
    if (use_frames) {
        Mat src;
        if (!imreadGray(frames_path[0].string(), src))
            throw std::runtime_error("cannot read frame " + frames_path[0].string());
        width = src.cols;
        height = src.rows;
        frames_num = (int)frames_path.size();
    } else {
        width = video_stream.width();
        height = video_stream.height();
        frames_num = video_stream.frameCount();
    }
    src_w_ = width; // with the resize on the device the loader queues SOURCE-size frames: buffers are sized from these
    src_h_ = height;
    bool do_resize = true;
    if (new_width > 0 && new_height > 0) {
        new_size = Size(new_width, new_height);
    } else if (new_width > 0) {
        new_size = Size(new_width, (int)std::round(height * 1.0 / width * new_width));
    } else if (new_height > 0) {
        new_size = Size((int)std::round(width * 1.0 / height * new_height), new_height);
    } else if (new_short > 0 && std::min(width, height) > new_short) {
        if (width < height)
            new_size = Size(new_short, (int)std::round(height * 1.0 / width * new_short));
        else
            new_size = Size((int)std::round(width * 1.0 / height * new_short), new_short);
    } else {
        do_resize = false;
        new_size = Size(width, height); // callers size their buffers from it
    }
    return do_resize;
}

// ------------------------------------------------------------------------------------------------ loading

static vector<path> list_frames(const path &dir) {
    vector<path> out;
    for (directory_iterator it(dir), end; it != end; ++it) {
        const string ext = it->path().extension().string();
        if (!is_regular_file(it->status()) || (ext != ".pgm" && ext != ".ppm"))
            continue; // the reference takes .jpg here (:249); this build has no JPEG decoder
        out.push_back(it->path());
    }
    std::sort(out.begin(), out.end());
    return out;
}

bool DenseFlow::load_frames_batch(VideoCapture &video_stream, const vector<path> &frames_path, bool use_frames,
                                  vector<Mat> &frames_gray, bool do_resize, const Size &size, bool to_gray) {
    // to_gray: the flow pipeline (sources are gray already: Y plane / PGM / BGR2GRAY in imreadGray); false: -s=0
    int cnt = 0;
    // A video with large frames starts with a quarter buffer, then a half one: the GPU begins after 64 frames instead
    // of 258 and is not left waiting for the first full buffer (the loader is only ~15 % faster than Farneback at 1080p:
    // profiles/round5/e2e/).  Buffer boundaries do not change any flow.
    const int limit = ramp_buffers_ > 0 ? std::max(std::min(32, batch_maxsize), batch_maxsize / (ramp_buffers_ == 2 ? 4 : 2))
                                        : batch_maxsize;
    if (ramp_buffers_ > 0)
        --ramp_buffers_;
    while (cnt < limit) {
        Mat frame;
        if (frames_budget == 0) // this pipeline's shard of the video ends here
            return false;
        if (use_frames) {
            if (cnt == (int)frames_path.size())
                return false;
            if (!imreadGray(frames_path[cnt].string(), frame))
                throw std::runtime_error("cannot read frame " + frames_path[cnt].string());
        } else if (!video_stream.read(frame)) {
            return false;
        }
        if (do_resize && !(to_gray && device_resize)) { // flow pipeline: the GPU resizes (calc_optflows_imp)
            Mat resized;
            resizeLinear(frame, resized, size);
            frames_gray.push_back(resized);
        } else {
            frames_gray.push_back(frame);
        }
        cnt++;
        if (frames_budget > 0)
            --frames_budget;
    }
    return true;
}

void DenseFlow::shard_range(int n_frames, int step, int rank, int world, int &begin, int &end) {
    const int m = std::max(n_frames - std::abs(step), 0);
    const int base = m / world, extra = m % world;
    begin = rank * base + std::min(rank, extra);
    end = begin + base + (rank < extra ? 1 : 0);
}

int DenseFlow::load_frames_video(VideoCapture &video_stream, vector<path> &frames_path, bool use_frames,
                                 bool do_resize, const Size &size, path output_dir, bool is_last, bool verbose) {
    int video_flow_idx = flow_begin_; // 0 unless this pipeline works on a shard of the video
    const int astep = std::abs(step);
    vector<Mat> padding;
    while (true) {
        vector<Mat> frames_gray;
        bool is_open;
        {
            StageTick tick(stage_us.load_read);
            is_open = load_frames_batch(video_stream, frames_path, use_frames, frames_gray, do_resize, size, true);
        }
        vector<Mat> padded(padding);
        padded.insert(padded.end(), frames_gray.begin(), frames_gray.end());
        if (verbose)
            cout << "push frames gray, video_flow_idx " << video_flow_idx << ", batch_size " << frames_gray.size()
                 << endl;
        TRACE("load: push %zu frames, base %d, last_buffer %d, final %d", padded.size(), video_flow_idx, (int)!is_open,
              (int)(is_last && !is_open));
        {
            StageTick tick(stage_us.load_push);
            frames_gray_queue.push(FlowBuffer(padded, output_dir, video_flow_idx, !is_open, false,
                                              (do_resize && device_resize) ? size : Size()),
                                   is_last && !is_open);
        }
        // the last |step| frames are needed again as the head of the next buffer (:204-207)
        padding.assign(padded.end() - std::min<size_t>(astep, padded.size()), padded.end());
        const int M = (int)padded.size() - astep;
        video_flow_idx += M;
        if (!is_open)
            break;
        // Drop the paths that were read.  (The reference erases M = read - |step| after the first buffer,
        // :213-215, so an image directory with more than one buffer of frames gets |step| frames twice;
        // deliberately not reproduced.)
        if (use_frames)
            frames_path.erase(frames_path.begin(), frames_path.begin() + std::min(frames_gray.size(), frames_path.size()));
    }
    return video_flow_idx - flow_begin_ + astep;
}

void DenseFlow::load_frames(bool use_frames, string save_type, bool verbose) {
    bool final_pushed = false;
    for (size_t i = 0; i < video_paths.size(); i++) {
        const path video_path = video_paths[i];
        const path output_dir = output_dirs[i];
        if (save_type == "h5") // create <output_dir>.h5 (reference :223-243); encode_save appends per FlowBuffer
            createHDF5(output_dir.string(), step);
        VideoCapture video_stream;
        vector<path> frames_path;
        if (use_frames) {
            frames_path = list_frames(video_path);
            if (frames_path.empty()) {
                if (verbose)
                    cout << video_path << " is empty!" << endl;
                continue;
            }
        } else if (!video_stream.open(video_path.string())) {
            throw std::runtime_error("cannot open video_path stream:" + video_path.string());
        }
        Size size;
        int frames_num;
        const bool do_resize = get_new_size(video_stream, frames_path, use_frames, size, frames_num);
        flow_begin_ = 0;
        frames_budget = -1;
        if (shard_world > 1) { // Level-2 sharding: only this pipeline's contiguous range of the video's flows
            if (frames_num < 0)
                throw std::runtime_error("cannot shard a stream of unknown length: " + video_path.string());
            int fb, fe;
            shard_range(frames_num, step, shard_rank, shard_world, fb, fe);
            const bool is_last_video = i == video_paths.size() - 1;
            if (fe == fb) { // fewer flows than shards: nothing to do here
                if (is_last_video) {
                    frames_gray_queue.push(FlowBuffer({}, path(), 0, false), true);
                    final_pushed = true;
                }
                continue;
            }
            flow_begin_ = fb;
            frames_budget = (long)(fe - fb) + std::abs(step);
            if (use_frames)
                frames_path.erase(frames_path.begin(), frames_path.begin() + fb);
            else if (!video_stream.seekFrame(fb))
                throw std::runtime_error("cannot seek in " + video_path.string());
        }
        // Frames per FlowBuffer.  The reference fixes 512 (include/dense_flow.h:33).  Here a buffer of large frames is
        // TWO FULL device batches (the engine advances min(512, 256 Mi / pixels) pairs together: 129 at 1080p, where a
        // batch of 64 runs 4 % slower): 258 frames at 1080p, 512 from 1024x1024 down — with -st=jpg the results are JPEG
        // files or 8-bit planes, so a buffer's memory is its frames.  The three stages only overlap across buffers, so the
        // FIRST buffer of a video is a quarter of that and the second a half (the GPU starts after 64 frames, not 258, and
        // does not run dry while the first full buffer is read).  Buffer boundaries do not change any flow (the last |step|
        // frames are carried over, :204-207).
        const long long frame_px = std::max((long long)size.width * size.height, (long long)src_w_ * src_h_);
        // (float flows — png / h5 / host-side bounding — are 8 bytes per pixel of page-locked output per pair: one batch)
        const long long budget = (device_bounding || device_png) ? (512ll << 20) : (256ll << 20);
        batch_maxsize = (int)std::max<long long>(32, std::min<long long>(512, budget / std::max(1ll, frame_px)));
        ramp_buffers_ = frame_px * batch_maxsize > (128ll << 20) ? 2 : 0; // only where reading a full buffer takes a while
        if (const char *bm = std::getenv("DF_BATCH_MAXSIZE")) // testing aid: force short buffers
            batch_maxsize = std::max(1, std::atoi(bm));
        if (verbose)
            cout << video_path << ", frames ≈ " << frames_num << endl;
        {   // early notice of the flows' size: the flow stage creates its engine while the first buffer is being read
            FlowBuffer hint({}, path(), 0, false);
            hint.engine_hint = size;
            frames_gray_queue.push(std::move(hint), false);
        }
        const bool is_last = i == video_paths.size() - 1;
        frames_num = load_frames_video(video_stream, frames_path, use_frames, do_resize, size, output_dir, is_last,
                                       verbose);
        final_pushed = is_last;
        total_frames += frames_num;
        if (verbose)
            cout << "loaded video " << video_path << ", " << frames_num << " frames" << endl;
    }
    if (!final_pushed) // e.g. the last input was empty: still let the downstream stages finish
        frames_gray_queue.push(FlowBuffer({}, path(), 0, false), true);
    if (verbose)
        cout << "load frames exit." << endl;
}

// ------------------------------------------------------------------------------------------------ the hot path

// Submitted FlowBuffers are collected by their own thread: it waits for a buffer's tail (dfx_wait, the one entry point
// that may run beside a submit) and hands the flows to the save stage the moment they are complete, in order.
void DenseFlow::collect_flows() {
    while (true) {
        std::unique_ptr<PendingFlows> p;
        {
            unique_lock<mutex> lock(pending_mtx_);
            pending_cv_.wait(lock, [&] { return !pending_q_.empty() || pending_closed_; });
            if (pending_q_.empty())
                return;
            p = std::move(pending_q_.front());
            pending_q_.pop();
        }
        bool failed;
        {
            unique_lock<mutex> lock(pending_mtx_);
            failed = !pending_error_.empty();
        }
        int wrc = DFX_OK;
        if (p->ticket && p->handle) {
            StageTick tick(stage_us.collect_wait);
            wrc = dfx_wait(p->handle, p->ticket);
        }
        if (wrc != DFX_OK && !failed) {
            const string msg = dfx_last_error(p->handle);
            unique_lock<mutex> lock(pending_mtx_);
            pending_error_ = msg.empty() ? string("dfx_wait failed") : msg;
            failed = true;
        }
        const bool fin = p->is_final;
        if (failed) {
            // The planes of this FlowBuffer (and of every later one) are undefined: nothing of them may reach the save
            // stage — it would write garbage files and, with is_record, mark the video done.  Closing the queues lets the
            // other stages wind down on an empty, unmarked final buffer; calc_optflows throws at its next call.
            frames_gray_queue.close();
            flows_queue.close();
        } else {
            size_t off = 0; // device JPEG: every FlowBuffer of the group gets its slice of the file sizes
            for (FlowBuffer &fb : p->flows)
                if (fb.encoded) {
                    const size_t m = fb.encoded->size_x.size();
                    std::copy(p->size_x.begin() + off, p->size_x.begin() + off + m, fb.encoded->size_x.begin());
                    std::copy(p->size_y.begin() + off, p->size_y.begin() + off + m, fb.encoded->size_y.begin());
                    off += m;
                }
            StageTick tick(stage_us.collect_push);
            for (size_t g = 0; g < p->flows.size(); ++g)
                flows_queue.push(std::move(p->flows[g]), fin && g + 1 == p->flows.size());
        }
        {
            unique_lock<mutex> lock(pending_mtx_);
            --pending_inflight_;
            pending_cv_.notify_all();
        }
        if (fin)
            return;
    }
}

// Every submitted FlowBuffer has reached the save stage (needed before the engine handle is destroyed).
void DenseFlow::flush_pending() {
    unique_lock<mutex> lock(pending_mtx_);
    pending_cv_.wait(lock, [&] { return pending_inflight_ == 0; });
    if (!pending_error_.empty())
        throw std::runtime_error(pending_error_);
}

void DenseFlow::enqueue_pending(std::unique_ptr<PendingFlows> p) {
    unique_lock<mutex> lock(pending_mtx_);
    // at most two FlowBuffers of flows wait for their tails: the one just submitted and the one before it
    pending_cv_.wait(lock, [&] { return pending_inflight_ < 2; });
    if (!pending_error_.empty()) // an earlier FlowBuffer's tail failed: stop at once, nothing more is saved
        throw std::runtime_error(pending_error_);
    ++pending_inflight_;
    pending_q_.push(std::move(p));
    pending_cv_.notify_all();
}

// The engine for flows of size sz (reference: the create() calls of :299-303, per FlowBuffer there).
void DenseFlow::prepare_engine(const string &algorithm, const Size &sz) {
    dfx_algo algo;
    const int rc = dfx_algo_from_name(algorithm.c_str(), &algo);
    if (rc != DFX_OK) { // "NV hardware flow not enabled, pls recompile" / "unknown optical algorithm <a>"
        char msg[256];
        throw std::runtime_error(dfx_algo_error_message(rc, algorithm.c_str(), msg, sizeof msg));
    }
    if (dfx_ && sz == dfx_size_)
        return;
    flush_pending();
    if (dfx_)
        dfx_destroy(dfx_);
    dfx_ = nullptr;
    // The reference's create() defaults, plus one engine knob: with several pipelines on one host (-g: one per GPU) the
    // threads that wait for the device — this stage inside dfx_submit_*, the collector in dfx_wait — sleep instead of
    // spinning: 1.6 instead of 2.5 (Farneback) / 4.1 (TVL1) CPU-ms per 1080p pair, CPUs the loader / encoder threads of
    // eight pipelines need on a 16-CPU allowance.  A single pipeline keeps the spinning waits: blocking costs Farneback
    // 15 % of its end-to-end rate (wake-up latency per FlowBuffer), TVL1 nothing (profiles/round4/e2e/).
    // DF_BLOCKING_WAIT=1 / DF_SPIN_WAIT=1 force either (A/B runs).
    dfx_params prm;
    dfx_default_params(&prm);
    prm.blocking_sync = std::getenv("DF_SPIN_WAIT") ? 0 : (std::getenv("DF_BLOCKING_WAIT") ? 1 : (blocking_waits_ ? 1 : 0));
    if (dfx_create(&dfx_, device, algo, sz.width, sz.height, &prm) != DFX_OK)
        throw std::runtime_error(dfx_last_error(nullptr));
    dfx_size_ = sz;
    TRACE("calc: engine for %dx%d ready", sz.width, sz.height);
}

// One FlowBuffer of gray frames -> M = max(N - |step|, 0) flows, on the GPU (reference :282-370).  The `stream`
// argument is the reference's (include/dense_flow.h:58-59); the engine owns its HIP streams, so it is only a tag here.
// The FlowBuffer is SUBMITTED: its last download overlaps the next FlowBuffer's uploads and compute, and its flows are
// pushed to the save stage when the next call (or the end of input) collects them.
void DenseFlow::calc_optflows_imp(const FlowBuffer &frames_gray, const string &algorithm, int step, bool verbose,
                                  Stream &stream) {
    (void)stream;
    vector<FlowBuffer> group(1, frames_gray);
    submit_group(group, algorithm, step, verbose);
}

// calc_optflows_imp for a GROUP of FlowBuffers of one geometry (normally one).  Several short clips are one library call
// (dfx_next_segments: pairs inside every clip only) so that the device batches are fuller than one clip's pairs; every
// FlowBuffer still gets its own result, in order, with its own output_dir / base_start / last_buffer.
void DenseFlow::submit_group(vector<FlowBuffer> &group, const string &algorithm, int step, bool verbose) {
    const bool is_final = flows_final_;
    const int astep = std::abs(step);
    vector<int> seg, m_of;
    int N = 0, M = 0;
    const Mat *first = nullptr;
    for (const FlowBuffer &fb : group) {
        const int n = (int)fb.item_data.size();
        seg.push_back(n);
        m_of.push_back(std::max(n - astep, 0));
        N += n, M += m_of.back();
        if (!first && n > 0)
            first = &fb.item_data[0];
    }
    std::unique_ptr<PendingFlows> pend(new PendingFlows());
    pend->is_final = is_final;
    vector<vector<Mat>> flows(group.size());
    vector<std::shared_ptr<FlowBuffer::Encoded>> encoded(group.size());
    vector<vector<double>> png_bounds(group.size());
    if (M > 0) {
        const Size in_sz = first->size();
        Size target;
        for (const FlowBuffer &fb : group)
            if (!fb.item_data.empty())
                target = fb.target;
        const Size sz = target.width > 0 ? target : in_sz; // size of the flows
        // One pitch and one source format describe the whole call: every frame must have the first one's geometry (the
        // reference resizes frame by frame, :166-170, so mixed-size image directories work there; here they would be
        // read out of bounds).
        vector<const uint8_t *> in;
        in.reserve(N);
        for (const FlowBuffer &fb : group)
            for (const Mat &f : fb.item_data) {
                if (!(f.size() == in_sz) || f.step != first->step || f.type() != CV_8UC1)
                    throw std::runtime_error("frames of one FlowBuffer differ in size (" + std::to_string(f.cols) + "x" +
                                             std::to_string(f.rows) + " after " + std::to_string(in_sz.width) + "x" +
                                             std::to_string(in_sz.height) + ")");
                in.push_back(f.ptr<uint8_t>());
            }
        TRACE("calc: %d frames of %d FlowBuffer(s) -> %d flows, %dx%d, algorithm %s", N, (int)group.size(), M, sz.width,
              sz.height, algorithm.c_str());
        prepare_engine(algorithm, sz); // sized per video; reused across its FlowBuffers (usually created already: engine_hint)
        // cv::resize of load_frames_batch (:169) on the device: source-size frames go up, the engine resizes
        if (dfx_set_source_format(dfx_, sz == in_sz ? 0 : in_sz.width, sz == in_sz ? 0 : in_sz.height, 1) != DFX_OK)
            throw std::runtime_error(dfx_last_error(dfx_));
        auto declare = [&] { // more than one clip in this call: pairs never cross a clip boundary
            if (group.size() > 1 && dfx_next_segments(dfx_, seg.data(), (int)seg.size()) != DFX_OK)
                throw std::runtime_error(dfx_last_error(dfx_));
        };
        uint64_t ticket = 0;
        bool encoded_on_device = false;
        if (device_jpeg) {
            // encodeFlowMap as a whole (src/common.cpp:48-64) happens on the device: convertFlowToImage(-bound, bound)
            // and both imencode(".jpg") — complete files come back, ~0.1 of the planes' bytes
            const size_t cap = dfx_jpeg_capacity(dfx_);
            pend->size_x.assign(M, 0u), pend->size_y.assign(M, 0u);
            vector<uint8_t *> out_x, out_y;
            for (size_t g = 0; g < group.size(); ++g) {
                if (m_of[g] == 0)
                    continue;
                encoded[g] = std::make_shared<FlowBuffer::Encoded>();
                encoded[g]->size_x.assign(m_of[g], 0u), encoded[g]->size_y.assign(m_of[g], 0u);
                for (int i = 0; i < m_of[g]; ++i) {
                    encoded[g]->x.emplace_back(new uchar[cap]);
                    encoded[g]->y.emplace_back(new uchar[cap]);
                    out_x.push_back(encoded[g]->x[i].get());
                    out_y.push_back(encoded[g]->y[i].get());
                }
            }
            declare();
            const int jrc = dfx_submit_batch_jpeg(dfx_, in.data(), first->step, N, step, -bound, bound,
                                                  95 /* cv::imencode's default quality */, out_x.data(), out_y.data(), cap,
                                                  pend->size_x.data(), pend->size_y.data(), &ticket);
            if (jrc == DFX_OK) {
                encoded_on_device = true;
            } else if (jrc != DFX_ERR_UNSUPPORTED) { // UNSUPPORTED: planes that do not compress; encode them on the host
                throw std::runtime_error(dfx_last_error(dfx_));
            } else {
                for (auto &e : encoded)
                    e.reset();
                pend->size_x.clear(), pend->size_y.clear();
            }
        }
        if (encoded_on_device) {
        } else if (device_png) {
            // convertFlowToPngImage's arithmetic (src/common.cpp:18-46) happens on the device: two 8-bit planes scaled by
            // the flow's own adaptive bounds come back with those bounds — 2 bytes per pixel instead of 8
            vector<uint8_t *> out_x, out_y;
            size_t out_step = 0;
            for (size_t g = 0; g < group.size(); ++g) {
                flows[g].resize(2 * (size_t)m_of[g]);
                for (int i = 0; i < m_of[g]; ++i) {
                    flows[g][2 * i].create(sz, CV_8UC1);
                    flows[g][2 * i + 1].create(sz, CV_8UC1);
                    out_x.push_back(flows[g][2 * i].ptr<uint8_t>());
                    out_y.push_back(flows[g][2 * i + 1].ptr<uint8_t>());
                    out_step = flows[g][2 * i].step;
                }
            }
            vector<double> all(2 * (size_t)M, 0.0); // complete when the submit call returns (include/dfx.h)
            declare();
            if (dfx_submit_batch_png(dfx_, in.data(), first->step, N, step, out_x.data(), out_y.data(), out_step, all.data(),
                                     &ticket) != DFX_OK)
                throw std::runtime_error(dfx_last_error(dfx_));
            size_t off = 0;
            for (size_t g = 0; g < group.size(); ++g) {
                png_bounds[g].assign(all.begin() + off, all.begin() + off + 2 * (size_t)m_of[g]);
                off += 2 * (size_t)m_of[g];
            }
        } else if (device_bounding) {
            // encodeFlowMap's convertFlowToImage(-bound, bound) (src/common.cpp:52) happens on the device:
            // two 8-bit planes per flow come back instead of a float field
            vector<uint8_t *> out_x, out_y;
            size_t out_step = 0;
            for (size_t g = 0; g < group.size(); ++g) {
                flows[g].resize(2 * (size_t)m_of[g]);
                for (int i = 0; i < m_of[g]; ++i) {
                    flows[g][2 * i].create(sz, CV_8UC1);
                    flows[g][2 * i + 1].create(sz, CV_8UC1);
                    out_x.push_back(flows[g][2 * i].ptr<uint8_t>());
                    out_y.push_back(flows[g][2 * i + 1].ptr<uint8_t>());
                    out_step = flows[g][2 * i].step;
                }
            }
            declare();
            if (dfx_submit_batch_u8(dfx_, in.data(), first->step, N, step, -bound, bound, out_x.data(), out_y.data(),
                                    out_step, &ticket) != DFX_OK)
                throw std::runtime_error(dfx_last_error(dfx_));
        } else {
            vector<float *> out;
            size_t out_step = 0;
            for (size_t g = 0; g < group.size(); ++g) {
                flows[g].resize(m_of[g]);
                for (int i = 0; i < m_of[g]; ++i) {
                    flows[g][i].create(sz, CV_32FC2);
                    out.push_back(flows[g][i].ptr<float>());
                    out_step = flows[g][i].step;
                }
            }
            declare();
            if (dfx_submit_batch(dfx_, in.data(), first->step, N, step, out.data(), out_step, &ticket) != DFX_OK)
                throw std::runtime_error(dfx_last_error(dfx_));
        }
        TRACE("calc: FlowBuffer submitted, ticket %llu", (unsigned long long)ticket);
        total_flows += M;
        pend->ticket = ticket;
        pend->handle = ticket ? dfx_ : nullptr;
    }
    if (verbose)
        cout << "flows queue push a item" << endl;
    // the collector thread waits for the tail (last download + hand-over) and pushes the flows to the save stage
    // while this thread already submits the next FlowBuffer
    for (size_t g = 0; g < group.size(); ++g) {
        FlowBuffer result(std::move(flows[g]), group[g].output_dir, group[g].base_start, group[g].last_buffer,
                          (device_bounding || device_png) && m_of[g] > 0);
        result.encoded = encoded[g];
        result.png_bounds = std::move(png_bounds[g]);
        pend->flows.push_back(std::move(result));
    }
    enqueue_pending(std::move(pend));
    // DF_SYNC_FLOW=1 (A/B measurements): collect every FlowBuffer at once, like the synchronous dfx_calc_batch*
    static const bool sync_flow = std::getenv("DF_SYNC_FLOW") != nullptr;
    if (sync_flow)
        flush_pending();
}

void DenseFlow::calc_optflows(bool verbose) {
    {
        unique_lock<mutex> lock(pending_mtx_);
        pending_closed_ = false;
    }
    thread collector([this] { collect_flows(); });
    std::exception_ptr err;
    try {
        // Joining: a FlowBuffer of at most 64 Mpx of frames may take along the FlowBuffers of the same geometry that are
        // ALREADY queued (nothing waits for a clip that has not been loaded yet), up to 256 Mpx per group.
        const size_t kJoinEach = 64u << 20, kJoinTotal = 256u << 20;
        auto frame_px = [](const FlowBuffer &b) {
            return b.item_data.empty() ? (size_t)0 : b.item_data.size() * (size_t)b.item_data[0].rows * b.item_data[0].cols;
        };
        auto same_geometry = [](const FlowBuffer &a, const FlowBuffer &b) {
            const Mat &x = a.item_data[0], &y = b.item_data[0];
            return x.size() == y.size() && x.step == y.step && x.type() == y.type() && a.target == b.target;
        };
        std::unique_ptr<std::pair<FlowBuffer, bool>> carry; // popped while joining, but it does not fit the group
        while (true) {
            bool is_final = false;
            vector<FlowBuffer> group;
            if (carry) {
                group.push_back(std::move(carry->first));
                is_final = carry->second;
                carry.reset();
            } else {
                StageTick tick(stage_us.flow_pop);
                group.push_back(frames_gray_queue.pop(&is_final));
            }
            if (group[0].engine_hint.width > 0) { // the loader's early notice: allocate while it reads the first frames
                flows_final_ = is_final;
                prepare_engine(algorithm, group[0].engine_hint);
                continue;
            }
            size_t px = frame_px(group[0]);
            while (join_short_ && !is_final && !group[0].item_data.empty() && px <= kJoinEach && group.size() < 64) {
                FlowBuffer next({}, path(), 0, false);
                bool fin = false;
                if (!frames_gray_queue.try_pop(next, &fin))
                    break;
                if (next.engine_hint.width > 0 && !fin) {
                    if (dfx_ && next.engine_hint == dfx_size_)
                        continue; // the next video has this group's size: its engine exists
                    carry.reset(new std::pair<FlowBuffer, bool>(std::move(next), fin));
                    break;
                }
                const size_t npx = frame_px(next);
                if (!next.item_data.empty() && (!same_geometry(group[0], next) || px + npx > kJoinTotal)) {
                    carry.reset(new std::pair<FlowBuffer, bool>(std::move(next), fin));
                    break;
                }
                px += npx;
                group.push_back(std::move(next));
                is_final = fin;
            }
            flows_final_ = is_final;
            {
                StageTick tick(stage_us.flow_submit);
                submit_group(group, algorithm, step, false);
            }
            if (is_final)
                break;
        }
        flush_pending();
    } catch (...) {
        err = std::current_exception();
    }
    {
        unique_lock<mutex> lock(pending_mtx_);
        pending_closed_ = true;
        pending_cv_.notify_all();
    }
    collector.join();
    if (err)
        std::rethrow_exception(err);
    if (verbose)
        cout << "calc optflows exit." << endl;
}

// ------------------------------------------------------------------------------------------------ encode + save

// `.done/<stem>` marker + the "done video" line (reference :456-470)
static void mark_done(const path &output_dir, bool has_class) {
    path donedir, title;
    if (has_class) {
        donedir = output_dir.parent_path().parent_path() / ".done" / output_dir.parent_path().filename();
        title = output_dir.parent_path().filename() / output_dir.filename();
    } else {
        donedir = output_dir.parent_path() / ".done";
        title = output_dir.filename();
    }
    createFile(donedir / output_dir.stem().string());
    cout << "done video " << title << endl;
}

void DenseFlow::encode_save(string save_type, bool verbose) {
    while (true) {
        bool is_final = false;
        FlowBuffer flow_buffer = [&] {
            StageTick tick(stage_us.save_pop);
            return flows_queue.pop(&is_final);
        }();
        StageTick work_tick(stage_us.save_work);
        const int M = flow_buffer.encoded ? (int)flow_buffer.encoded->x.size()
                                          : (int)flow_buffer.item_data.size() / (flow_buffer.bounded ? 2 : 1);
        TRACE("save: %d flows, base %d, final %d", M, flow_buffer.base_start, (int)is_final);
        // The flows of a buffer are independent: the encoders run encode_threads wide (one thread in the
        // reference, :414-437).  Files are still written in index order by this thread.
        if (save_type == "jpg" && flow_buffer.encoded) {
            // complete JPEG files arrive from the device: nothing left to encode, and nothing to copy — the files are
            // written straight from the buffers.  (Eight writers instead of one were measured: no faster within the
            // run-to-run noise on a list of 224x224 clips, and twice the CPU time — 6.3 instead of 3.1 ms per 1080p
            // pair, the threads contend inside the file system: profiles/round3/experiments/e2e_parallel_writers.log.)
            const FlowBuffer::Encoded &e = *flow_buffer.encoded;
            const string px = (flow_buffer.output_dir / "flow_x").string(), py = (flow_buffer.output_dir / "flow_y").string();
            for (int i = 0; i < M; ++i)
                writeFlowImageBytes(e.x[i].get(), e.size_x[i], px, step, flow_buffer.base_start + i);
            for (int i = 0; i < M; ++i)
                writeFlowImageBytes(e.y[i].get(), e.size_y[i], py, step, flow_buffer.base_start + i);
            TRACE("save: written");
        } else if (save_type == "jpg") {
            vector<vector<uchar>> output_x(M), output_y(M);
            if (flow_buffer.bounded) { // planes arrive bounded from the device: encode only
                parallelFor(2 * M, encode_threads, [&](int k) {
                    if (!imencodeJpeg(flow_buffer.item_data[k], (k & 1) ? output_y[k / 2] : output_x[k / 2]))
                        throw std::runtime_error("JPEG encoder failed");
                });
            } else {
                parallelFor(M, encode_threads, [&](int i) {
                    Mat planes[2];
                    split(flow_buffer.item_data[i], planes);
                    encodeFlowMap(planes[0], planes[1], output_x[i], output_y[i], bound);
                });
            }
            TRACE("save: encoded");
            writeFlowImages(output_x, (flow_buffer.output_dir / "flow_x").string(), step, flow_buffer.base_start);
            writeFlowImages(output_y, (flow_buffer.output_dir / "flow_y").string(), step, flow_buffer.base_start);
            TRACE("save: written");
        } else if (save_type == "png") {
            vector<vector<uchar>> output(M);
            if (flow_buffer.bounded) { // the scheme's planes and bounds arrive from the device: interleave + encode only
                parallelFor(M, encode_threads, [&](int i) {
                    encodeFlowMapPngPlanes(flow_buffer.item_data[2 * i], flow_buffer.item_data[2 * i + 1],
                                           flow_buffer.png_bounds[2 * i], flow_buffer.png_bounds[2 * i + 1], output[i]);
                });
            } else {
                parallelFor(M, encode_threads, [&](int i) {
                    Mat planes[2];
                    split(flow_buffer.item_data[i], planes);
                    encodeFlowMapPng(planes[0], planes[1], output[i]);
                });
            }
            writeFlowImagesPng(output, (flow_buffer.output_dir / "flow").string(), step, flow_buffer.base_start);
        } else if (save_type == "h5") { // unbounded float planes (reference :429-441)
            vector<Mat> output_h5_x(M), output_h5_y(M);
            parallelFor(M, encode_threads, [&](int i) {
                Mat planes[2];
                split(flow_buffer.item_data[i], planes);
                output_h5_x[i] = planes[0];
                output_h5_y[i] = planes[1];
            });
            writeHDF5(output_h5_x, flow_buffer.output_dir.string(), "flow_x", step, flow_buffer.base_start);
            writeHDF5(output_h5_y, flow_buffer.output_dir.string(), "flow_y", step, flow_buffer.base_start);
        }
        // mark the video done after its last buffer has been written (resume support, :456-470)
        if (is_record && flow_buffer.last_buffer)
            mark_done(flow_buffer.output_dir, has_class);
        if (is_final)
            break;
    }
    if (verbose)
        cout << "post process exit." << endl;
}

// ------------------------------------------------------------------------------------------------ frames only (-s=0)

int DenseFlow::extract_frames_video(VideoCapture &video_stream, vector<path> &frames_path, bool use_frames,
                                    bool do_resize, const Size &size, path output_dir, bool verbose) {
    (void)verbose;
    int video_frame_idx = 0;
    while (true) {
        vector<Mat> frames;
        const bool is_open = load_frames_batch(video_stream, frames_path, use_frames, frames, do_resize, size, false);
        vector<vector<uchar>> output_img;
        for (const Mat &f : frames) {
            vector<uchar> str_img;
            imencodeJpeg(f, str_img);
            output_img.push_back(std::move(str_img));
        }
        writeImages(output_img, (output_dir / "img").string(), video_frame_idx);
        video_frame_idx += (int)frames.size();
        if (!is_open)
            break;
        if (use_frames)
            frames_path.erase(frames_path.begin(), frames_path.begin() + frames.size());
    }
    return video_frame_idx;
}

void DenseFlow::extract_frames_only(bool use_frames, bool verbose) {
    // Deliberate limitation, stated loudly: the reference's -s=0 mode writes COLOUR frames (it reads BGR,
    // src/denseflow_gpu.cpp:100-117); this decoder-free build has a gray-only JPEG encoder and reads the Y plane of
    // .y4m clips / converts .ppm to gray, so the extracted img_%05d.jpg are single-channel.
    static std::once_flag warned;
    std::call_once(warned, [] {
        cout << "note: -s=0 in this build writes GRAY frames (Y plane / BGR2GRAY); the reference writes colour frames"
             << endl;
    });
    for (size_t i = 0; i < video_paths.size(); i++) {
        VideoCapture video_stream;
        vector<path> frames_path;
        if (use_frames) {
            frames_path = list_frames(video_paths[i]);
            if (frames_path.empty()) {
                if (verbose)
                    cout << video_paths[i] << " is empty!" << endl;
                continue;
            }
        } else if (!video_stream.open(video_paths[i].string())) {
            throw std::runtime_error("cannot open video_path stream:" + video_paths[i].string());
        }
        Size size;
        int frames_num;
        const bool do_resize = get_new_size(video_stream, frames_path, use_frames, size, frames_num);
        frames_num = extract_frames_video(video_stream, frames_path, use_frames, do_resize, size, output_dirs[i], verbose);
        total_frames += frames_num;
        if (verbose)
            cout << "extracted frames of video " << video_paths[i] << ", " << frames_num << " frames" << endl;
    }
}

// ------------------------------------------------------------------------------------------------ entry points

void DenseFlow::launch(bool use_frames, string save_type, bool verbose) {
    std::exception_ptr err[3];
    auto guarded = [&](int k, auto fn) {
        return [&, k, fn] {
            try {
                fn();
            } catch (...) { // the reference lets worker exceptions hit std::terminate; report them instead
                err[k] = std::current_exception();
                frames_gray_queue.close(); // unblock the other stages; they wind down on an empty final buffer
                flows_queue.close();
            }
        };
    };
    thread t_load(guarded(0, [&] { load_frames(use_frames, save_type, verbose); }));
    thread t_calc(guarded(1, [&] { calc_optflows(false); }));
    thread t_save(guarded(2, [&] { encode_save(save_type, false); }));
    t_load.join();
    TRACE("launch: load joined");
    t_calc.join();
    TRACE("launch: calc joined");
    t_save.join();
    TRACE("launch: save joined");
    if (g_trace || std::getenv("DF_STAGES")) { // where each stage's wall time went (one line per pipeline)
        const StageTimes &t = stage_us;
        fprintf(stderr,
                "[denseflow stages, device %d, ms] load: read %.0f / blocked on the frames queue %.0f | flow: waiting for "
                "frames %.0f / in the library %.0f | collect: dfx_wait %.0f / blocked on the flows queue %.0f | save: waiting "
                "for flows %.0f / encode + write %.0f\n",
                device, t.load_read / 1e3, t.load_push / 1e3, t.flow_pop / 1e3, t.flow_submit / 1e3, t.collect_wait / 1e3,
                t.collect_push / 1e3, t.save_pop / 1e3, t.save_work / 1e3);
    }
    for (auto &e : err)
        if (e)
            std::rethrow_exception(e);
}

vector<Mat> DenseFlowTestAccess::run_calc_optflows_imp(DenseFlow &d, const vector<Mat> &frames_gray,
                                                        const string &algorithm, int step, bool bounded) {
    d.device_bounding = bounded;
    d.device_jpeg = false; // this probe returns flows / bounded planes; the encoded form is tested through the CLI
    d.device_png = false;
    d.flows_final_ = true;
    thread collector([&d] { d.collect_flows(); });
    std::exception_ptr err;
    try {
        d.calc_optflows_imp(FlowBuffer(frames_gray, path(), 0, true), algorithm, step, false, d.stream);
        d.flush_pending();
    } catch (...) {
        err = std::current_exception();
        unique_lock<mutex> lock(d.pending_mtx_);
        d.pending_closed_ = true;
        d.pending_cv_.notify_all();
    }
    collector.join();
    if (err)
        std::rethrow_exception(err);
    bool fin = false;
    return d.flows_queue.pop(&fin).item_data;
}

static void print_summary(size_t n_videos, unsigned long N, unsigned long F, const string &algorithm, double secs) {
    cout << n_videos << " videos (" << N << " frames, " << F << " " << algorithm << " flows) processed, using " << secs
         << "s, decoding speed " << N / secs << "fps, flow speed " << F / secs << "fps" << endl;
}

void calcDenseFlowVideoGPU(vector<path> video_paths, vector<path> output_dirs, string algorithm, int step, int bound,
                           int new_width, int new_height, int new_short, bool has_class, bool use_frames,
                           string save_type, bool is_record, bool verbose) {
    calcDenseFlowVideoMultiGPU(video_paths, output_dirs, algorithm, step, bound, new_width, new_height, new_short,
                               has_class, use_frames, save_type, is_record, verbose, vector<int>{0});
}

void calcDenseFlowVideoMultiGPU(vector<path> video_paths, vector<path> output_dirs, string algorithm, int step,
                                int bound, int new_width, int new_height, int new_short, bool has_class,
                                bool use_frames, string save_type, bool is_record, bool verbose, vector<int> devices) {
    if (devices.empty())
        devices.push_back(0);
    // Level 1: videos are dealt round-robin to the devices, each device runs its own three-thread
    // pipeline.  Level 2: with fewer videos than devices (one long clip on an 8-GPU node) every pipeline takes a
    // contiguous range of EVERY video's flows instead; file indices stay global, the `.done` marker is written here
    // once all shards of a video are on disk.  (-st=h5 keeps Level 1: one writer per file.)
    const bool split = step != 0 && save_type != "h5" && !video_paths.empty() && video_paths.size() < devices.size();
    const size_t G = split ? devices.size() : std::min(devices.size(), std::max<size_t>(video_paths.size(), 1));
    vector<std::unique_ptr<DenseFlow>> workers;
    for (size_t g = 0; g < G; ++g) {
        vector<path> vp, od;
        for (size_t i = split ? 0 : g; i < video_paths.size(); i += split ? 1 : G) {
            vp.push_back(video_paths[i]);
            od.push_back(output_dirs[i]);
        }
        workers.emplace_back(new DenseFlow(vp, od, algorithm, step, bound, new_width, new_height, new_short, has_class,
                                           is_record && !split, save_type, devices[g]));
        if (split)
            workers.back()->set_shard((int)g, (int)G);
        workers.back()->set_blocking_waits(G > 1);
    }
    const double start_t = CurrentSeconds();
    // DF_PROCESSES=1: one PROCESS per pipeline instead of one thread set (Level 1 / Level 2 sharding unchanged).  Two
    // processes on ONE device (DF_DEVICES=0,0) overlap their launch gaps and host phases where two handles of one process
    // share a HIP runtime and do not (round 2: 403 -> 430 pairs/s for TVL1 at 1080p); it is also the rehearsal of an
    // N-rank launch on a one-GPU box.  The parent must not have touched HIP before the fork (tools/denseflow.cpp takes the
    // device list from DF_DEVICES then, or counts the devices in a child of their own), builds nothing itself and only adds
    // up the children's counts.
    static const bool use_processes = std::getenv("DF_PROCESSES") != nullptr;
    unsigned long N = 0, F = 0;
    if (use_processes && G > 1) {
        std::cout.flush();
        std::cerr.flush();
        vector<pid_t> pids(G);
        vector<int> fds(G);
        for (size_t g = 0; g < G; ++g) {
            int pp[2];
            if (pipe(pp) != 0)
                throw std::runtime_error("pipe() failed");
            const pid_t pid = fork();
            if (pid < 0)
                throw std::runtime_error("fork() failed");
            if (pid == 0) { // child: its own pipeline, its own HIP runtime
                close(pp[0]);
                for (size_t k = 0; k < g; ++k) // the read ends of the children started before this one
                    close(fds[k]);
                int rc = 0;
                try {
                    if (step == 0)
                        workers[g]->extract_frames_only(use_frames, verbose);
                    else
                        workers[g]->launch(use_frames, save_type, verbose);
                    const unsigned long cnt[2] = {workers[g]->get_processed_total_frames(), workers[g]->get_processed_total_flows()};
                    if (write(pp[1], cnt, sizeof cnt) != (ssize_t)sizeof cnt)
                        rc = 1;
                } catch (const std::exception &ex) {
                    cout << ex.what() << endl;
                    rc = 1;
                } catch (...) { // whatever it is, the child must reach _exit and never run the parent's code below
                    rc = 1;
                }
                std::cout.flush();
                std::cerr.flush();
                _exit(rc); // no destructors of the parent's copies, no atexit handlers of a runtime this process did not start
            }
            close(pp[1]);
            pids[g] = pid, fds[g] = pp[0];
        }
        bool failed = false;
        for (size_t g = 0; g < G; ++g) {
            unsigned long cnt[2] = {0, 0};
            const bool got = read(fds[g], cnt, sizeof cnt) == (ssize_t)sizeof cnt;
            close(fds[g]);
            int status = 0;
            waitpid(pids[g], &status, 0);
            if (!got || !WIFEXITED(status) || WEXITSTATUS(status) != 0)
                failed = true;
            N += cnt[0], F += cnt[1];
        }
        if (failed)
            throw std::runtime_error("a pipeline process failed (its message is above)");
    } else {
    vector<std::exception_ptr> errs(G);
    vector<thread> threads;
    for (size_t g = 0; g < G; ++g)
        threads.emplace_back([&, g] {
            try {
                if (step == 0)
                    workers[g]->extract_frames_only(use_frames, verbose);
                else
                    workers[g]->launch(use_frames, save_type, verbose);
            } catch (...) {
                errs[g] = std::current_exception();
            }
        });
    for (auto &t : threads)
        t.join();
    for (auto &e : errs)
        if (e)
            std::rethrow_exception(e);
    for (auto &w : workers) {
        N += w->get_processed_total_frames(); // split: the |step| overlap frames are counted by every shard
        F += w->get_processed_total_flows();
    }
    }
    if (split && is_record)
        for (const path &od : output_dirs)
            mark_done(od, has_class);
    const double end_t = CurrentSeconds();
    print_summary(video_paths.size(), N, F, algorithm, std::max(end_t - start_t, 1e-3));
}
