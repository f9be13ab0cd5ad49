// valu_rate.hip — issue-rate probe for the VALU instructions the exact TVL1 arithmetic is made of (round 4).
// One workgroup per CU x `waves` waves per SIMD, each wave runs N iterations of 8 independent instructions of one kind;
// prints cycles per wave-instruction per SIMD.  Measurement tooling (hipcc --offload-arch=gfx950 -O3 valu_rate.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND> __global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int n) {
    float f[8]; double d[8]; typedef float f2 __attribute__((ext_vector_type(2))); f2 p[8];
    for (int i = 0; i < 8; ++i) { f[i] = 1.0f + threadIdx.x * 1e-3f + i; d[i] = 1.0 + threadIdx.x * 1e-3 + i; p[i].x = f[i]; p[i].y = f[i] + 0.5f; }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));
#define RSQ32(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
#define RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
#define RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define SQRT32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
#define CVT6432(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define CVT3264(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define ADD32(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[i]));
#define MAX32(i) asm volatile("v_max_f32 %0, %0, %0" : "+v"(f[i]));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(f[i]));
        if (KIND == 0) { REP8(FMA32) } else if (KIND == 1) { REP8(PKFMA) } else if (KIND == 2) { REP8(PKADD) }
        else if (KIND == 3) { REP8(FMA64) } else if (KIND == 4) { REP8(MUL64) } else if (KIND == 5) { REP8(ADD64) }
        else if (KIND == 6) { REP8(RSQ32) } else if (KIND == 7) { REP8(RSQ64) } else if (KIND == 8) { REP8(RCP32) }
        else if (KIND == 9) { REP8(CVT6432) } else if (KIND == 10) { REP8(CVT3264) } else if (KIND == 11) { REP8(ADD32) }
        else if (KIND == 12) { REP8(PKMUL) } else if (KIND == 13) { REP8(SQRT32) } else if (KIND == 14) { REP8(MAX32) }
        else if (KIND == 15) { REP8(CNDMASK) }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0; for (int i = 0; i < 8; ++i) acc += f[i] + (float)d[i] + p[i].x + p[i].y;
    if (acc == 123.456f) out[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> double run(const char *name, int threads, int n) {
    float *out; long long *cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v; s /= 256;
    const double waves_per_simd = threads / 64.0 / 4.0;
    const double per = s / ((double)n * 8 * waves_per_simd); // cycles per wave-instruction per SIMD
    printf("%-14s threads=%4d  %.2f cycles per wave-instruction per SIMD\n", name, threads, per);
    hipFree(out); hipFree(cyc); return per;
}

int main() {
    const int n = 4000;
    for (int threads : {256, 1024}) {
        run<0>("v_fma_f32", threads, n); run<11>("v_add_f32", threads, n); run<14>("v_max_f32", threads, n); run<15>("v_cndmask_b32", threads, n);
        run<1>("v_pk_fma_f32", threads, n); run<2>("v_pk_add_f32", threads, n); run<12>("v_pk_mul_f32", threads, n);
        run<3>("v_fma_f64", threads, n); run<4>("v_mul_f64", threads, n); run<5>("v_add_f64", threads, n);
        run<6>("v_rsq_f32", threads, n); run<8>("v_rcp_f32", threads, n); run<13>("v_sqrt_f32", threads, n); run<7>("v_rsq_f64", threads, n);
        run<9>("v_cvt_f64_f32", threads, n); run<10>("v_cvt_f32_f64", threads, n);
    }
    return 0;
}
