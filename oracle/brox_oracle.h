/*
 * oracle/brox_oracle.h — TEST INFRASTRUCTURE ONLY (see oracle_common.h header).
 *
 * CPU restatement of cv::cuda::BroxOpticalFlow as the reference calls it:
 *   create(0.197f, 50.0f, 0.8f, 10, 77, 10)            /root/reference/src/denseflow_gpu.cpp:303
 *   convertTo(CV_32F, 1.0/255.0) on both frames, calc()  /root/reference/src/denseflow_gpu.cpp:332-334
 * Upstream: opencv_contrib 4.5.2 cudaoptflow/src/brox.cpp -> cudalegacy NCVBroxOpticalFlow.cu (+ NPPST
 * resize / separable filters).  None of it is under /root/reference and the reference has no golden
 * data for it.  PARITY UNPINNED, and SURVEY.md Appendix C rates the recollection of upstream details LOW:
 * this header therefore DEFINES the algorithm (structure as upstream, every free choice listed).
 *
 * Definition (Brox/Bruhn/Papenberg/Weickert 2004, NCV structure):
 *  1. I = (float)u8 * (float)(1.0/255.0).
 *  2. Pyramid: level 0 = frame.  While prev_w > 15 && prev_h > 15 && levels < outer_iterations:
 *     scale *= scale_factor (float); w = ceilf(W*scale), h = ceilf(H*scale); the level is the PREVIOUS
 *     level down-sampled by area averaging ("supersample": box [x*s, x*s + s) with s = 1/scale_factor,
 *     fractional weights at the box ends, source index clamped, sum / weight-sum).
 *  3. u = v = 0 at the coarsest level.  Per level, coarse to fine:
 *     a. derivatives with the 5-tap filter (1,-8,0,8,-1)/12, mirror border with edge duplication
 *        (index -1 -> 0, w -> w-1):  Ix0,Iy0 of I0;  Ix,Iy of I1;  Ixx = Dx(Ix), Iyy = Dy(Iy), Ixy = Dy(Ix).
 *     b. du = dv = 0; inner_iterations times:
 *        stage 1 per pixel: sample I1,Ix,Iy,Ixx,Ixy,Iyy bilinearly (exact float weights, mirror indices) at
 *          (x + u, y + v); Iz = I1w - I0, Ixz = Ixw - Ix0, Iyz = Iyw - Iy0;
 *          q0 = Iz + Ixw du + Iyw dv, q1 = Ixz + Ixxw du + Ixyw dv, q2 = Iyz + Ixyw du + Iyyw dv;
 *          psi = 0.5f * (1/sqrtf(q0^2 + gamma (q1^2 + q2^2) + 1e-6f)) / alpha;
 *          num_dudv = psi (Ixw Iyw + gamma (Ixxw Ixyw + Ixyw Iyyw)),
 *          den_u = psi (Ixw^2 + gamma (Ixyw^2 + Ixxw^2)), den_v = psi (Iyw^2 + gamma (Ixyw^2 + Iyyw^2)),
 *          num_u = psi (Ixw Iz + gamma (Ixxw Ixz + Ixyw Iyz)), num_v = psi (Iyw Iz + gamma (Iyyw Iyz + Ixyw Ixz));
 *          diffusivities of w = (u+du, v+dv) on the staggered grid, neighbours replicated at the border:
 *            gx(x,y) [between x-1 and x]: wx = w(x,y)-w(x-1,y), wy = 0.25 (w(x,y+1)+w(x-1,y+1)-w(x,y-1)-w(x-1,y-1))
 *            gy(x,y) [between y-1 and y]: wy = w(x,y)-w(x,y-1), wx = 0.25 (w(x+1,y)+w(x+1,y-1)-w(x-1,y)-w(x-1,y-1))
 *            g = 0.5f * (1/sqrtf(ux^2 + uy^2 + vx^2 + vy^2 + 1e-6f));  gx(0,y) = gy(x,0) = 0 (Neumann).
 *        stage 2: inv_den_u = 1/(den_u + gx(x,y) + gx(x+1,y) + gy(x,y) + gy(x,y+1)), same for v
 *          (g taken as 0 beyond the last column/row).
 *        solver_iterations times: red pass ((x+y) even) then black pass, omega = 1.99f:
 *          su = gl (u_l+du_l) + gr (u_r+du_r) + gd (u_d+du_d) + gu (u_u+du_u) - (gl+gr+gd+gu) u
 *          du' = (1-omega) du + omega inv_den_u (su - num_u - num_dudv dv)
 *          dv' = (1-omega) dv + omega inv_den_v (sv - num_v - num_dudv du')     (Gauss-Seidel in the 2x2 coupling:
 *                the freshly updated du' is used; with the old du the iteration diverges at omega = 1.99)
 *     c. u += du, v += dv.  If a finer level exists: u,v = bicubic resize (Catmull-Rom A = -0.5, source
 *        coordinate x*(w_coarse/w_fine)... = x * scale_factor, window [ceil(x-2), floor(x+2)] clipped to the
 *        image, sum/weight-sum) * (1/scale_factor).
 *  4. flow = (u, v) of level 0, interleaved.
 * float32 throughout, no FMA contraction, 1/sqrtf and divisions IEEE.
 */
#ifndef DFX_BROX_ORACLE_H
#define DFX_BROX_ORACLE_H

#include "oracle_common.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float alpha, gamma, scale_factor;
    int inner_iterations, outer_iterations, solver_iterations;
} orc_brox_params;

void orc_brox_default_params(orc_brox_params *p); /* the literals of src/denseflow_gpu.cpp:303 */

/* levels[] receives the pyramid sizes (w,h pairs, finest first); returns the number of levels */
int orc_brox_pyramid_sizes(int W, int H, const orc_brox_params *p, int *wh, int max_levels);

int orc_brox_calc(const uint8_t *I0, size_t pitch0, const uint8_t *I1, size_t pitch1, int W, int H,
                  const orc_brox_params *params, float *flow_uv);

/* stage functions used by unit tests */
void orc_brox_downsample(const float *src, int sw, int sh, float *dst, int dw, int dh, float factor);
void orc_brox_upsample_bicubic(const float *src, int sw, int sh, float *dst, int dw, int dh, float factor, float mul);
void orc_brox_deriv_x(const float *src, int w, int h, float *dst);
void orc_brox_deriv_y(const float *src, int w, int h, float *dst);

#ifdef __cplusplus
}
#endif
#endif
