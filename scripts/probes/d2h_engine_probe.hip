// d2h_engine_probe.hip — which engine executes a device-to-host hipMemcpy on this stack?  One variant per run; run under
// `rocprofv3 --kernel-trace --memory-copy-trace`: a shader blit shows up as __amd_rocclr_copyBuffer dispatches in the
// kernel trace, an SDMA transfer as a MEMORY_COPY_DEVICE_TO_HOST record (scripts/r3_gpu11.sh counts both).
//   hipcc --offload-arch=gfx950 -O2 -o build/d2h_engine_probe scripts/probes/d2h_engine_probe.hip
//   build/d2h_engine_probe <variant>   0 default pinned | 1 non-coherent | 2 coherent | 3 portable+mapped | 4 2-D copy |
//                                      5 hipMemcpyDtoHAsync | 6 hipHostRegister'ed malloc | 7 write-combined | 8 numa-user |
//                                      9 every copy waits for an event of a kernel on another stream | 10 a kernel runs
//                                      on another stream meanwhile (no dependency) | 11 copies on the NULL stream
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

__global__ void k_spin(float *p, int iters) {
    float a = p[threadIdx.x];
    for (int i = 0; i < iters; ++i)
        a = a * 1.0001f + 0.5f;
    p[threadIdx.x] = a;
}

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t flow = (size_t)1920 * 1080 * 8, n = 32, bytes = flow * n;
    CK(hipSetDevice(0));
    char *d = nullptr, *h = nullptr;
    CK(hipMalloc((void **)&d, bytes));
    CK(hipMemset(d, 1, bytes));
    unsigned flags = hipHostMallocDefault;
    if (variant == 1) flags = hipHostMallocNonCoherent;
    if (variant == 2) flags = hipHostMallocCoherent;
    if (variant == 3) flags = hipHostMallocPortable | hipHostMallocMapped;
    if (variant == 7) flags = hipHostMallocWriteCombined;
    if (variant == 8) flags = hipHostMallocNumaUser;
    if (variant == 6) {
        h = (char *)aligned_alloc(4096, bytes);
        for (size_t i = 0; i < bytes; i += 4096) h[i] = 0;
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
    } else {
        CK(hipHostMalloc((void **)&h, bytes, flags));
    }
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    float *dk = nullptr;
    CK(hipMalloc((void **)&dk, 4096));
    CK(hipMemset(dk, 0, 4096));
    if (variant == 11)
        s = nullptr;
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < n; ++k) {
        if (variant == 9 || variant == 10) {
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s2, dk, 20000);
            if (variant == 9) {
                CK(hipEventRecord(ev, s2));
                CK(hipStreamWaitEvent(s, ev, 0));
            }
        }
        if (variant == 4)
            CK(hipMemcpy2DAsync(h + k * flow, 1920 * 8, d + k * flow, 1920 * 8, 1920 * 8, 1080, hipMemcpyDeviceToHost, s));
        else if (variant == 5)
            CK(hipMemcpyDtoHAsync(h + k * flow, (hipDeviceptr_t)(d + k * flow), flow, s));
        else
            CK(hipMemcpyAsync(h + k * flow, d + k * flow, flow, hipMemcpyDeviceToHost, s));
    }
    CK(hipStreamSynchronize(s));
    CK(hipStreamSynchronize(s2));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("variant %d: %zu copies of %.1f MB device -> host: %.1f GB/s\n", variant, n, flow / 1e6, bytes / dt / 1e9);
    return 0;
}
