// dfx_prof — a torch-free, Python-free process around dfx_calc_batch_device, for rocprofv3.
//
// rocprofv3 --pmc on the torch-hosted bench.py segfaults at the bench's own batch on this pool
// (profiles/round3/pmc_batch129_attempt.log), so round 3's HBM-traffic figures were measured at batch 16 and scaled
// (VERDICT r3 weak #6).  This program needs nothing but the C ABI of include/dfx.h: frames from a raw file (u8, W x H x N,
// written by scripts/make_raw_clip.py from the bench's SynthClip), resident in HBM, `passes` timed passes of the whole
// FlowBuffer, one JSON line with the rate and the engine's own statistics.  Measurement tooling, not part of the product.
//
//   dfx_prof <algo tvl1|farn|brox> <W> <H> <frames.raw> <n_frames> <step> <passes> [max_batch] [variant] [tvl1_math] [block]
//            [tvl1_epsilon | -1 = default] [clips: the frames are `clips` clips of n_frames / clips frames, one FlowBuffer]
#include <sys/resource.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dfx.h"

static void die(const char *what, dfx_handle h = nullptr) {
    std::fprintf(stderr, "dfx_prof: %s%s%s\n", what, h ? ": " : "", h ? dfx_last_error(h) : "");
    std::exit(1);
}

int main(int argc, char **argv) {
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s algo W H frames.raw n_frames step passes [max_batch] [variant] [tvl1_math] [blocking_sync]\n",
                     argv[0]);
        return 2;
    }
    dfx_algo algo;
    if (dfx_algo_from_name(argv[1], &algo) != DFX_OK)
        die("unknown algorithm");
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[5]), step = std::atoi(argv[6]);
    const int passes = std::atoi(argv[7]);
    dfx_params prm;
    dfx_default_params(&prm);
    if (argc > 8)
        prm.max_batch = std::atoi(argv[8]);
    if (argc > 9)
        prm.variant = std::atoi(argv[9]);
    if (argc > 10)
        prm.tvl1_math = std::atoi(argv[10]);
    if (argc > 11)
        prm.blocking_sync = std::atoi(argv[11]);
    if (argc > 12 && std::atof(argv[12]) >= 0.0)
        prm.tvl1_epsilon = std::atof(argv[12]);
    const int clips = argc > 13 ? std::max(1, std::atoi(argv[13])) : 1;
    const size_t fbytes = (size_t)W * H;
    std::vector<uint8_t> frames(fbytes * N);
    FILE *f = std::fopen(argv[4], "rb");
    if (!f || std::fread(frames.data(), 1, frames.size(), f) != frames.size())
        die("cannot read the raw clip (u8, W*H*n_frames bytes)");
    std::fclose(f);

    dfx_handle h = nullptr;
    if (dfx_create(&h, 0, algo, W, H, &prm) != DFX_OK)
        die("dfx_create", h);
    if (N % clips != 0)
        die("n_frames must be a multiple of clips");
    const int per_clip = N / clips - std::abs(step) > 0 ? N / clips - std::abs(step) : 0;
    const int M = per_clip * clips;
    std::vector<int> seg((size_t)clips, N / clips);
    void *d_frames = nullptr, *d_flows = nullptr;
    if (dfx_device_malloc(h, &d_frames, frames.size()) != DFX_OK ||
        dfx_device_malloc(h, &d_flows, (size_t)M * fbytes * 8) != DFX_OK)
        die("device allocation", h);
    if (dfx_memcpy_h2d(h, d_frames, frames.data(), frames.size()) != DFX_OK)
        die("upload", h);
    auto pass = [&]() {
        if (clips > 1 && dfx_next_segments(h, seg.data(), clips) != DFX_OK)
            die("dfx_next_segments", h);
        if (dfx_calc_batch_device(h, (const uint8_t *)d_frames, (size_t)W, fbytes, N, step, (float *)d_flows,
                                  fbytes * 2) != DFX_OK)
            die("dfx_calc_batch_device", h);
    };
    pass(); // warm-up: allocations, first-touch
    dfx_reset_stats(h);
    auto cpu_s = []() { // user + system CPU seconds of this process (all threads)
        rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        return ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
    };
    const double c0 = cpu_s();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < passes; ++i)
        pass();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double cpu = cpu_s() - c0;
    dfx_stats st;
    dfx_get_stats(h, &st);
    // a checksum of the last flow so that two builds can be compared for identical output
    std::vector<float> last((size_t)fbytes * 2);
    unsigned long long sum = 0;
    if (M > 0) {
        if (dfx_memcpy_d2h(h, last.data(), (const char *)d_flows + (size_t)(M - 1) * fbytes * 8, fbytes * 8) != DFX_OK)
            die("download", h);
        for (size_t i = 0; i < last.size(); ++i) {
            unsigned u;
            std::memcpy(&u, &last[i], 4);
            sum = sum * 1000003ULL + u;
        }
    }
    std::printf("{\"algo\":\"%s\",\"W\":%d,\"H\":%d,\"frames\":%d,\"step\":%d,\"passes\":%d,\"pairs_per_s\":%.3f,"
                "\"batch\":%d,\"step_launches\":%llu,\"avg_launch_us\":%.3f,\"device_ms_per_pair\":%.5f,"
                "\"step_algorithmic_bytes_per_launch\":%.1f,\"kernel_launches\":%llu,\"noop_steps\":%llu,"
                "\"tvl1_mean_iters\":%.3f,\"cpu_ms_per_pair\":%.4f,\"cpu_busy_fraction\":%.3f,\"blocking_sync\":%d,"
                "\"last_flow_checksum\":\"%016llx\"}\n",
                argv[1], W, H, N, step, passes, passes * (double)M / dt, st.batch, (unsigned long long)st.step_launches,
                st.step_launches ? st.step_ms * 1e3 / (double)st.step_launches : 0.0,
                st.pairs ? st.device_ms / (double)st.pairs : 0.0,
                st.step_launches ? st.step_algorithmic_bytes / (double)st.step_launches : 0.0,
                (unsigned long long)st.kernel_launches, (unsigned long long)st.noop_steps,
                st.pairs ? (double)st.tvl1_total_iters / (double)st.pairs : 0.0,
                passes * M > 0 ? cpu * 1e3 / (passes * (double)M) : 0.0, cpu / dt, prm.blocking_sync, sum);
    dfx_device_free(h, d_frames);
    dfx_device_free(h, d_flows);
    dfx_destroy(h);
    return 0;
}
