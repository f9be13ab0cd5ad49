// pcie_probe.hip — host-link bandwidth of the box, measured with plain page-locked hipMemcpyAsync.
//   hipcc --offload-arch=gfx950 -O2 -o build/pcie_probe scripts/probes/pcie_probe.hip && build/pcie_probe
// Reference context: the blocking flow_gpu.download of /root/reference/src/denseflow_gpu.cpp:339 and the uploads at
// :317-318 — 16.6 MB of CV_32FC2 flow down and 2 x 2.07 MB of gray frames up per 1080p pair.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);                \
            exit(1);                                                                                   \
        }                                                                                              \
    } while (0)

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Lane {
    hipStream_t s;
    char *h, *d;
    size_t bytes;
};

// n copies of `chunk` bytes per lane, all lanes at once; dir[i]: 0 = H2D, 1 = D2H.  Returns seconds.
static double run(std::vector<Lane> &lanes, const std::vector<int> &dir, size_t chunk, int n) {
    for (auto &l : lanes)
        CK(hipStreamSynchronize(l.s));
    const double t0 = now();
    for (int k = 0; k < n; ++k)
        for (size_t i = 0; i < lanes.size(); ++i) {
            Lane &l = lanes[i];
            const size_t off = ((size_t)k * chunk) % (l.bytes - chunk + 1);
            if (dir[i])
                CK(hipMemcpyAsync(l.h + off, l.d + off, chunk, hipMemcpyDeviceToHost, l.s));
            else
                CK(hipMemcpyAsync(l.d + off, l.h + off, chunk, hipMemcpyHostToDevice, l.s));
        }
    for (auto &l : lanes)
        CK(hipStreamSynchronize(l.s));
    return now() - t0;
}

int main(int argc, char **argv) {
    const size_t big = (argc > 1 ? (size_t)atoll(argv[1]) : 2048) << 20; // MiB per lane buffer
    const unsigned flags = argc > 2 ? (unsigned)atoi(argv[2]) : hipHostMallocDefault;
    CK(hipSetDevice(0));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, pciBus %02x:%02x, hipHostMalloc flags %u, lane buffer %zu MiB\n", p.name, p.pciBusID,
           p.pciDeviceID, flags, big >> 20);
    std::vector<Lane> lanes(2);
    for (auto &l : lanes) {
        CK(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
        CK(hipHostMalloc((void **)&l.h, big, flags));
        CK(hipMalloc((void **)&l.d, big));
        memset(l.h, 1, big); // touch every page
        CK(hipMemset(l.d, 2, big));
        l.bytes = big;
    }
    CK(hipDeviceSynchronize());
    const size_t flow = (size_t)1920 * 1080 * 8, frame = (size_t)1920 * 1080;
    struct Case {
        const char *name;
        size_t chunk;
        int n;
    } cases[] = {{"one 1080p gray frame (2.07 MB)", frame, 512},
                 {"one 1080p CV_32FC2 flow (16.6 MB)", flow, 96},
                 {"256 MiB", (size_t)256 << 20, 8},
                 {"whole buffer", big, 2}};
    std::vector<Lane> one(lanes.begin(), lanes.begin() + 1);
    for (const Case &c : cases) {
        if (c.chunk > big)
            continue;
        run(one, {1}, c.chunk, 2);
        const double d2h = run(one, {1}, c.chunk, c.n);
        run(one, {0}, c.chunk, 2);
        const double h2d = run(one, {0}, c.chunk, c.n);
        const double both = run(lanes, {0, 1}, c.chunk, c.n);
        const double d2h2 = run(lanes, {1, 1}, c.chunk, c.n);
        const double gb = (double)c.chunk * c.n / 1e9;
        printf("%-36s x%4d : D2H %6.1f GB/s | H2D %6.1f GB/s | H2D+D2H at once %6.1f + %6.1f GB/s | two D2H streams "
               "%6.1f GB/s total\n",
               c.name, c.n, gb / d2h, gb / h2d, gb / both, gb / both, 2 * gb / d2h2);
    }
    // the Farneback f32-out pattern: per pair one 16.6 MB flow down while two frames go up
    {
        const int n = 96;
        for (auto &l : lanes)
            CK(hipStreamSynchronize(l.s));
        const double t0 = now();
        for (int k = 0; k < n; ++k) {
            CK(hipMemcpyAsync(lanes[1].h + (size_t)(k % 64) * flow, lanes[1].d + (size_t)(k % 64) * flow, flow,
                              hipMemcpyDeviceToHost, lanes[1].s));
            CK(hipMemcpyAsync(lanes[0].d + (size_t)(k % 64) * frame, lanes[0].h + (size_t)(k % 64) * frame, frame,
                              hipMemcpyHostToDevice, lanes[0].s));
        }
        for (auto &l : lanes)
            CK(hipStreamSynchronize(l.s));
        const double dt = now() - t0;
        printf("flow down (16.6 MB) + frame up (2.07 MB) per pair, two streams: %.0f pairs/s = %.1f GB/s down\n", n / dt,
               n * (double)flow / dt / 1e9);
    }
    // 2-D copy with a row pitch (what a non-contiguous cv::Mat needs): 1080 rows of 15360 B into rows 16384 B apart
    {
        const int n = 48;
        const size_t pitch = 16384;
        CK(hipStreamSynchronize(lanes[0].s));
        const double t0 = now();
        for (int k = 0; k < n; ++k)
            CK(hipMemcpy2DAsync(lanes[0].h, pitch, lanes[0].d, 1920 * 8, 1920 * 8, 1080, hipMemcpyDeviceToHost,
                                lanes[0].s));
        CK(hipStreamSynchronize(lanes[0].s));
        const double dt = now() - t0;
        printf("hipMemcpy2DAsync D2H of one flow into pitched rows: %.1f GB/s\n", n * (double)flow / dt / 1e9);
    }
    return 0;
}
