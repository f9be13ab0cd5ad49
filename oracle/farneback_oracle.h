/*
 * oracle/farneback_oracle.h — TEST INFRASTRUCTURE ONLY (see oracle_common.h header).
 *
 * CPU restatement of cv::cuda::FarnebackOpticalFlow (opencv_contrib 4.5.2,
 * modules/cudaoptflow/src/farneback.cpp + src/cuda/farneback.cu) as called by the reference at
 * /root/reference/src/denseflow_gpu.cpp:301 (create(), defaults) and :329 (calc).
 * Specification followed: SURVEY.md Appendix B (+ Appendix E helpers).
 * PARITY UNPINNED (no reference golden vectors exist; OpenCV unavailable here).
 */
#ifndef DFX_FARNEBACK_ORACLE_H
#define DFX_FARNEBACK_ORACLE_H

#include "oracle_common.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int num_levels;
    double pyr_scale;
    int fast_pyramids; /* must be 0 */
    int win_size;
    int num_iters;
    int poly_n; /* 5 or 7 */
    double poly_sigma;
    int flags; /* must be 0 (box-filter update path) */
} orc_farneback_params;

typedef struct {
    float g[8], xg[8], xxg[8]; /* index 0..poly_n (the device uses the non-negative half) */
    float ig11, ig03, ig33, ig55;
} orc_farneback_poly_consts;

void orc_farneback_default_params(orc_farneback_params *p);

int orc_farneback_calc(const uint8_t *I0, size_t pitch0, const uint8_t *I1, size_t pitch1, int W, int H,
                       const orc_farneback_params *params, float *flow_uv);

/* stage functions (dense planes) */
void orc_farneback_prepare_poly(int n, double sigma, orc_farneback_poly_consts *out);            /* B.3 */
int orc_farneback_gaussian_kernel(int ksize, double sigma, float *k /* ksize floats */);         /* B.6 */
void orc_farneback_gaussian_blur(const float *src, int W, int H, const float *ker_half, int half, float *dst); /* B.4 */
void orc_farneback_poly_exp(const float *src, int W, int H, int n, const orc_farneback_poly_consts *c,
                            float *R /* 5 planes */);                                             /* B.5 */
void orc_farneback_update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1, int W,
                                   int H, float *M /* 5 planes */);                               /* B.7 */
void orc_farneback_box_filter5(const float *src, int W, int H, int half, float *dst);             /* B.8 */
void orc_farneback_update_flow(const float *M, int W, int H, float *flowx, float *flowy);         /* B.9 */

#ifdef __cplusplus
}
#endif
#endif
