/*
 * oracle/tvl1_oracle.h — TEST INFRASTRUCTURE ONLY (see oracle_common.h header).
 *
 * CPU restatement of cv::cuda::OpticalFlowDual_TVL1 (opencv_contrib 4.5.2,
 * modules/cudaoptflow/src/tvl1flow.cpp + src/cuda/tvl1flow.cu) as called by the
 * reference at /root/reference/src/denseflow_gpu.cpp:299 (create(), defaults)
 * and :327 (calc).  Specification followed: SURVEY.md Appendix A.
 * PARITY UNPINNED (no reference golden vectors exist; OpenCV unavailable here).
 */
#ifndef DFX_TVL1_ORACLE_H
#define DFX_TVL1_ORACLE_H

#include "oracle_common.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SCALES 16
#define ORC_MAX_WARPS 16
#define ORC_MAX_CHECKS 8192

typedef struct {
    double tau, lambda, theta;
    int nscales, warps;
    double epsilon;
    int iterations;
    double scale_step;
    double gamma; /* must be 0: the u3/p3 path is not restated (denseflow never enables it) */
} orc_tvl1_params;

typedef struct {
    int nscales; /* levels actually used after the <16 px crop (A.2 step 3) */
    int w[ORC_MAX_SCALES], h[ORC_MAX_SCALES];
    int iters[ORC_MAX_SCALES][ORC_MAX_WARPS]; /* inner iterations executed */
    int n_checks;                             /* convergence checks performed (all counted, first ORC_MAX_CHECKS logged) */
    int chk_level[ORC_MAX_CHECKS], chk_warp[ORC_MAX_CHECKS], chk_n[ORC_MAX_CHECKS];
    double chk_err[ORC_MAX_CHECKS];
} orc_tvl1_trace;

void orc_tvl1_default_params(orc_tvl1_params *p);

/* Full calc(): two 8-bit frames -> interleaved (u,v) float flow, H*W*2. trace may be NULL. */
int orc_tvl1_calc(const uint8_t *I0, size_t pitch0, const uint8_t *I1, size_t pitch1, int W, int H,
                  const orc_tvl1_params *params, float *flow_uv, orc_tvl1_trace *trace);

/* Stage functions (A.3, A.5-A.7), dense planes of W*H floats. */
void orc_tvl1_centered_gradient(const float *I1, int W, int H, float *I1x, float *I1y);
void orc_tvl1_warp_backward(const float *I0, const float *I1, const float *I1x, const float *I1y, const float *u1,
                            const float *u2, int W, int H, float *I1w, float *I1wx, float *I1wy, float *grad,
                            float *rho_c);
/* returns sum(diff) in double if calc_error, else 0 */
double orc_tvl1_estimate_u(const float *I1wx, const float *I1wy, const float *grad, const float *rho_c,
                           const float *p11, const float *p12, const float *p21, const float *p22, float *u1,
                           float *u2, int W, int H, float l_t, float theta, int calc_error);
void orc_tvl1_estimate_dual(const float *u1, const float *u2, float *p11, float *p12, float *p21, float *p22, int W,
                            int H, float taut);
/* procOneScale (A.3): u1,u2 in/out. level = index used for trace bookkeeping. */
void orc_tvl1_proc_one_scale(const float *I0, const float *I1, float *u1, float *u2, int W, int H,
                             const orc_tvl1_params *params, int level, orc_tvl1_trace *trace);

#ifdef __cplusplus
}
#endif
#endif
