/*
 * oracle/cpu_tvl1_baseline.h — TEST / BENCH INFRASTRUCTURE ONLY (see oracle_common.h header).
 *
 * The CPU comparator BASELINE.json's north_star names: "the reference's OpenCV CPU TVL1 timed on the
 * same box's host cores".  A denseflow user without a GPU would swap the cv::cuda::OpticalFlowDual_TVL1
 * calls of /root/reference/src/denseflow_gpu.cpp:299,327 for cv::optflow::DualTVL1OpticalFlow
 * (opencv_contrib 4.5.2, modules/optflow/src/tvl1flow.cpp).  OpenCV is not available in this
 * environment, so this is an OpenMP restatement ("port") of that CPU class with its create() defaults,
 * following SURVEY.md Appendix D: half-pixel-centre cv::resize pyramids, cv::remap(INTER_CUBIC, A = -0.75,
 * 1/32-px coordinates, BORDER_CONSTANT 0) warping, 10 outer x 30 inner iterations with the error evaluated
 * every iteration, medianBlur 5x5 of u1/u2 at every outer iteration, and upstream's separate passes with
 * temporaries (v, div_p, u_x, u_y).
 *
 * It is a TIMING baseline, not a parity oracle: CPU and CUDA OpenCV differ by far more than 1e-3
 * (SURVEY.md H1), the parity oracle is tvl1_oracle.c (cv::cuda semantics).  Unpinned like the rest.
 */
#ifndef DFX_CPU_TVL1_BASELINE_H
#define DFX_CPU_TVL1_BASELINE_H

#include "oracle_common.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CPU_TVL1_MAX_SCALES 16

typedef struct {
    double tau, lambda, theta;
    int nscales, warps;
    double epsilon;
    int inner_iterations, outer_iterations;
    double scale_step;
    int median_filtering; /* 1 = off, 3 or 5 */
} cpu_tvl1_params;

typedef struct {
    int nscales;
    int w[CPU_TVL1_MAX_SCALES], h[CPU_TVL1_MAX_SCALES];
    long long inner_iterations; /* executed, all levels and warps */
    long long outer_iterations;
    double px_iterations;       /* sum over levels of pixels x executed inner iterations */
} cpu_tvl1_stats;

void cpu_tvl1_default_params(cpu_tvl1_params *p);

/* calc(): two 8-bit gray frames -> interleaved (u, v) float flow (H*W*2).  stats may be NULL. */
int cpu_tvl1_calc(const uint8_t *I0, size_t pitch0, const uint8_t *I1, size_t pitch1, int W, int H,
                  const cpu_tvl1_params *params, float *flow_uv, cpu_tvl1_stats *stats);

/* building blocks, exposed for the known-answer tests */
void cpu_tvl1_resize_linear(const float *src, int sw, int sh, float *dst, int dw, int dh, double inv_scale_x,
                            double inv_scale_y);                                  /* cv::resize INTER_LINEAR, 32F */
void cpu_tvl1_median_blur(const float *src, float *dst, int w, int h, int ksize); /* cv::medianBlur 32F, 3 or 5  */
void cpu_tvl1_remap_cubic(const float *src, int w, int h, const float *mapx, const float *mapy,
                          float *dst);                                            /* cv::remap INTER_CUBIC       */
void cpu_tvl1_cubic_coeffs(float x, float coeffs[4]);                            /* interpolateCubic, A = -0.75 */

#ifdef __cplusplus
}
#endif
#endif
