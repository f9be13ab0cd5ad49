// include/dense_flow.h — the reference's public API (/root/reference/include/dense_flow.h:6-100) on top
// of the MI355X C ABI (include/dfx.h).  Same names, argument meaning and error behaviour:
//   calcDenseFlowVideoGPU(...)      reference :6-8,  body src/denseflow_gpu.cpp:479-497
//   class FlowBuffer                reference :10-18
//   class DenseFlow                 reference :20-100 (launch, extract_frames_only, get_processed_total_*)
// The only member that touches the GPU is calc_optflows_imp (reference :58-59, body
// src/denseflow_gpu.cpp:282-370); here it calls dfx_calc_batch instead of cv::cuda::*OpticalFlow.
#ifndef DENSEFLOW_DENSE_FLOW_H
#define DENSEFLOW_DENSE_FLOW_H

#include "common.h"
#include "image_io.h"

void calcDenseFlowVideoGPU(vector<path> video_paths, vector<path> output_dirs, string algorithm, int step, int bound,
                           int new_width, int new_height, int new_short, bool has_class, bool use_frames,
                           string save_type, bool is_record, bool verbose);

// Extension: the same job sharded over several GPUs of one node (videos round-robin, one DenseFlow and
// one handle per device, no collective).  devices = {0} is calcDenseFlowVideoGPU.
void calcDenseFlowVideoMultiGPU(vector<path> video_paths, vector<path> output_dirs, string algorithm, int step,
                                int bound, int new_width, int new_height, int new_short, bool has_class,
                                bool use_frames, string save_type, bool is_record, bool verbose, vector<int> devices);

class FlowBuffer {
  public:
    vector<Mat> item_data;
    path output_dir;
    int base_start;
    bool last_buffer;
    // Extension: flows already bounded to 8 bits on the device.  item_data then holds 2M CV_8UC1 planes,
    // x and y alternating (flow i = item_data[2i], item_data[2i+1]), instead of M CV_32FC2 fields.
    bool bounded;
    // Extension: flows already ENCODED on the device (dfx_submit_batch_jpeg): complete JPEG files, item_data is empty.
    // Plain (pageable, uninitialised) buffers: the library assembles the files on the host, nothing is DMA'd into them.
    struct Encoded {
        vector<std::unique_ptr<uchar[]>> x, y; // file i of flow_x / flow_y
        vector<uint32_t> size_x, size_y;       // its size in bytes (valid once the FlowBuffer's ticket has been waited on)
    };
    std::shared_ptr<Encoded> encoded;
    // Extension: the -st=png scheme's planes from the device (dfx_submit_batch_png): `bounded` planes as above, scaled by
    // the reference's per-flow adaptive bounds; png_bounds[2i], [2i+1] = bound_x, bound_y of flow i (empty otherwise).
    vector<double> png_bounds;
    // Extension: not a buffer of frames but the loader's early notice of the size the next video's flows will have
    // (width > 0): the flow stage creates its engine (device allocations, ~0.3 s at 1080p) while the loader reads the
    // first frames instead of after them.  Nothing is forwarded to the save stage.
    Size engine_hint;
    // Extension: frames that still have the source size; the flow stage resizes them to `target` on the device
    // (width 0: the frames already have their final size).
    Size target;
    FlowBuffer(vector<Mat> item_data, path output_dir, int base_start, bool last_buffer, bool bounded = false,
               Size target = Size())
        : item_data(std::move(item_data)), output_dir(std::move(output_dir)), base_start(base_start),
          last_buffer(last_buffer), bounded(bounded), target(target) {}
};

// Bounded producer/consumer queue (the reference hand-rolls two of these with a mutex and two
// condition variables each, include/dense_flow.h:35-47).
class FlowBufferQueue {
  public:
    explicit FlowBufferQueue(size_t maxsize) : maxsize_(maxsize) {}
    void push(FlowBuffer b, bool is_final);
    // pops one buffer; *was_final tells the consumer that no further buffer will ever arrive
    FlowBuffer pop(bool *was_final);
    // the same without waiting: false when nothing is queued right now
    bool try_pop(FlowBuffer &out, bool *was_final);
    // Short clips: the reference's bound of `maxsize` buffers starves a consumer that joins several of them into one
    // device batch.  With a byte budget the queue also accepts a buffer while fewer than `bytes` bytes (and fewer than
    // `hard_max` buffers) are queued; large FlowBuffers stay bounded by `maxsize` alone.
    void set_byte_budget(size_t bytes, size_t hard_max) { byte_budget_ = bytes, hard_max_ = hard_max; }
    // a stage died: producers stop blocking, consumers see an empty final buffer
    void close();
    size_t size();

  private:
    size_t maxsize_;
    size_t byte_budget_ = 0, hard_max_ = 0, bytes_ = 0;
    mutex mtx_;
    condition_variable not_full_, not_empty_;
    queue<std::pair<FlowBuffer, bool>> q_;
    queue<size_t> q_bytes_;
    bool closed_ = false;
};

class DenseFlow {
  private:
    vector<path> video_paths;
    vector<path> output_dirs;
    string algorithm;
    string save_type;
    int step;
    int bound;
    int new_width;
    int new_height;
    int new_short;
    bool has_class;
    bool is_record;
    int device;
    // Extensions of the save stage.  device_bounding: for save_type "jpg" the flows are
    // bounded to 8 bits on the GPU (dfx_calc_batch_u8) and the host only encodes; DF_HOST_BOUND=1 restores
    // the reference's host-side convertFlowToImage.  encode_threads: JPEG/PNG encoders running in parallel
    // inside encode_save (the reference encodes on one thread); DF_ENCODE_THREADS overrides.
    bool device_bounding;
    // device_jpeg: for save_type "jpg" the bounded planes are also JPEG-coded on the GPU (dfx_submit_batch_jpeg:
    // encodeFlowMap as a whole, src/common.cpp:48-64); the save stage only writes files.  DF_HOST_JPEG=1 keeps the
    // encoders on the host (device bounding only).
    bool device_jpeg;
    // device_png: for save_type "png" the arithmetic of convertFlowToPngImage (src/common.cpp:18-46: minMaxLoc, the
    // adaptive bounds, the two convertTo planes) runs on the GPU (dfx_submit_batch_png); the save stage interleaves
    // the planes with the bound channel and calls the PNG encoder.  DF_HOST_PNG=1 keeps the float path (A/B, tests).
    bool device_png;
    int encode_threads;
    // Extension of the load stage: a requested resize (-nw/-nh/-ns) is done by the flow stage
    // on the GPU (dfx_set_source_format) instead of cv::resize on the loader thread; DF_HOST_RESIZE=1 restores
    // the host-side resize (same integer arithmetic, same output).
    bool device_resize;

    int batch_maxsize;
    int ramp_buffers_ = 0; // the next FlowBuffers of a video with large frames: 2 -> a quarter, 1 -> half of batch_maxsize
    // Level-2 sharding: this pipeline computes flows [begin, end) of every video, the contiguous
    // range of shard `shard_rank` of `shard_world`; output indices stay global through base_start.  Frames
    // [begin, end + |step|) are loaded — the |step| overlap frames the reference's own batch padding duplicates
    // (src/denseflow_gpu.cpp:204-208).  world 1 = the whole video.
    int shard_rank = 0, shard_world = 1;
    long frames_budget = -1; // frames the loader may still read from the current video (-1: to its end)
    int flow_begin_ = 0;     // global index of the first flow this pipeline computes in the current video
    int src_w_ = 0, src_h_ = 0; // size of the current video's decoded frames (get_new_size)
    FlowBufferQueue frames_gray_queue;
    FlowBufferQueue flows_queue;
    unsigned long total_frames;
    unsigned long total_flows;
    // Where each stage's wall time goes, in microseconds (DF_TRACE / DF_STAGES print them after the run): reading /
    // decoding frames vs blocked on the full frames queue; waiting for frames vs inside the library call; waiting for
    // flows vs encoding + writing.  One writer per counter (its stage's thread).
    struct StageTimes {
        long long load_read = 0, load_push = 0, flow_pop = 0, flow_submit = 0, collect_wait = 0, collect_push = 0,
                  save_pop = 0, save_work = 0;
    } stage_us;

    // the device engine (replaces Ptr<cuda::*OpticalFlow> + cv::cuda::Stream)
    dfx_handle dfx_;
    Size dfx_size_;
    Stream stream; // reference member (include/dense_flow.h:33)
    // the FlowBuffer whose last download is still in flight (dfx_submit_batch*), and whether the buffer being
    // processed is the last of the run
    struct PendingFlows {
        vector<FlowBuffer> flows; // one result per FlowBuffer of the group that was submitted together, in order
        uint64_t ticket = 0;
        bool is_final = false;
        dfx_handle handle = nullptr; // the engine the ticket belongs to (nullptr: nothing to wait for)
        // device JPEG: the file sizes of the whole group, written by the library until the ticket is collected; the
        // collector hands every FlowBuffer its slice (FlowBuffer::Encoded::size_x / size_y)
        vector<uint32_t> size_x, size_y;
    };
    queue<std::unique_ptr<PendingFlows>> pending_q_; // submitted, tails not yet collected (collect_flows)
    mutex pending_mtx_;
    condition_variable pending_cv_;
    int pending_inflight_ = 0;
    bool pending_closed_ = false;
    string pending_error_;
    bool flows_final_ = false;
    // Short FlowBuffers of one geometry (a list of small clips, BASELINE configs[3]) that are already waiting are joined
    // into ONE library call (dfx_next_segments): the same flows, fuller device batches.  DF_NO_JOIN=1 disables it.
    bool join_short_ = true;
    bool blocking_waits_ = false; // sleep, not spin, while waiting for the device (set when several pipelines share the host)
    void submit_group(vector<FlowBuffer> &group, const string &algorithm, int step, bool verbose);
    void collect_flows();
    void enqueue_pending(std::unique_ptr<PendingFlows> p);

    bool check_param();
    bool get_new_size(const VideoCapture &video_stream, const vector<path> &frames_path, bool use_frames,
                      Size &new_size, int &frames_num);
    bool load_frames_batch(VideoCapture &video_stream, const vector<path> &frames_path, bool use_frames,
                           vector<Mat> &frames_gray, bool do_resize, const Size &size, bool to_gray);
    int load_frames_video(VideoCapture &video_stream, vector<path> &frames_path, bool use_frames, bool do_resize,
                          const Size &size, path output_dir, bool is_last, bool verbose);
    // the reference's signature (include/dense_flow.h:58-59); `stream` is a tag here, the engine owns its HIP streams
    void calc_optflows_imp(const FlowBuffer &frames_gray, const string &algorithm, int step, bool verbose,
                           Stream &stream = Stream::Null());
    void flush_pending();
    void prepare_engine(const string &algorithm, const Size &sz);
    void load_frames(bool use_frames, string save_type, bool verbose = true);
    void calc_optflows(bool verbose = true);
    void encode_save(string save_type, bool verbose = true);
    int extract_frames_video(VideoCapture &video_stream, vector<path> &frames_path, bool use_frames, bool do_resize,
                             const Size &size, path output_dir, bool verbose);
    friend struct DenseFlowTestAccess;

  public:
    void launch(bool use_frames, string save_type, bool verbose);
    void set_shard(int rank, int world) {
        shard_rank = rank;
        shard_world = world;
    }
    void set_blocking_waits(bool on) { blocking_waits_ = on; }
    // flows [begin, end) of a clip of n_frames frames that shard `rank` of `world` computes (contiguous, balanced)
    static void shard_range(int n_frames, int step, int rank, int world, int &begin, int &end);
    void extract_frames_only(bool use_frames, bool verbose);
    unsigned long get_processed_total_frames() { return total_frames; }
    unsigned long get_processed_total_flows() { return total_flows; }

    DenseFlow(vector<path> video_paths, vector<path> output_dirs, string algorithm, int step, int bound, int new_width,
              int new_height, int new_short, bool has_class, bool is_record, string save_type, int device = 0);
    ~DenseFlow();
    DenseFlow(const DenseFlow &) = delete;
    DenseFlow &operator=(const DenseFlow &) = delete;
};

// Test hook: drive the GPU operator on in-memory frames exactly as calc_optflows does.
struct DenseFlowTestAccess {
    // bounded = false: M CV_32FC2 flows; true: 2M CV_8UC1 planes bounded on the device (x, y alternating)
    static vector<Mat> run_calc_optflows_imp(DenseFlow &d, const vector<Mat> &frames_gray, const string &algorithm,
                                             int step, bool bounded = false);
};

#endif // DENSEFLOW_DENSE_FLOW_H
