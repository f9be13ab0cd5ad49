/*
 * oracle/cpu_tvl1_baseline.c — TEST / BENCH INFRASTRUCTURE ONLY.  See cpu_tvl1_baseline.h.
 *
 * OpenMP restatement of CPU cv::optflow::DualTVL1OpticalFlow (opencv_contrib 4.5.2
 * modules/optflow/src/tvl1flow.cpp: calc, procOneScale, and the *Body parallel loops), SURVEY.md
 * Appendix D.  This is what a CPU-only user of the reference would run instead of the cv::cuda call at
 * /root/reference/src/denseflow_gpu.cpp:299,327 — the timing comparator of BASELINE.json's north_star.
 * Every pass is a row-parallel loop like upstream's cv::parallel_for_ bodies.
 */
#include "cpu_tvl1_baseline.h"

#include <float.h>
#include <omp.h>

void cpu_tvl1_default_params(cpu_tvl1_params *p) {
    /* DualTVL1OpticalFlow::create() defaults */
    p->tau = 0.25;
    p->lambda = 0.15;
    p->theta = 0.3;
    p->nscales = 5;
    p->warps = 5;
    p->epsilon = 0.01;
    p->inner_iterations = 30;
    p->outer_iterations = 10;
    p->scale_step = 0.8;
    p->median_filtering = 5;
}

/* ---- cv::resize, INTER_LINEAR, CV_32F: half-pixel centres, clamped taps, horizontal then vertical ---- */
void cpu_tvl1_resize_linear(const float *src, int sw, int sh, float *dst, int dw, int dh, double inv_scale_x,
                            double inv_scale_y) {
    const double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    float *xa = (float *)malloc(sizeof(float) * 2 * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) {
            fx = 0.f;
            sx = 0;
        }
        if (sx >= sw - 1) {
            fx = 0.f;
            sx = sw - 1;
        }
        xofs[dx] = sx;
        xa[2 * dx] = 1.f - fx;
        xa[2 * dx + 1] = fx;
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        int sy0 = sy, sy1 = sy + 1;
        if (sy0 < 0)
            sy0 = 0;
        if (sy0 > sh - 1)
            sy0 = sh - 1;
        if (sy1 < 0)
            sy1 = 0;
        if (sy1 > sh - 1)
            sy1 = sh - 1;
        const float b0 = 1.f - fy, b1 = fy;
        const float *S0 = src + (size_t)sy0 * sw, *S1 = src + (size_t)sy1 * sw;
        float *D = dst + (size_t)dy * dw;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            const float a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
            const float r0 = S0[sx] * a0 + S0[sx1] * a1;
            const float r1 = S1[sx] * a0 + S1[sx1] * a1;
            D[dx] = r0 * b0 + r1 * b1;
        }
    }
    free(xofs);
    free(xa);
}

/* ---- cv::medianBlur on CV_32F (sorting-network path), BORDER_REPLICATE ---- */
#define CSWAP(a, b)                                                                                                  \
    do {                                                                                                             \
        const float lo_ = (a) < (b) ? (a) : (b);                                                                     \
        const float hi_ = (a) < (b) ? (b) : (a);                                                                     \
        (a) = lo_;                                                                                                   \
        (b) = hi_;                                                                                                   \
    } while (0)

static inline float median9(float *p) {
    CSWAP(p[1], p[2]); CSWAP(p[4], p[5]); CSWAP(p[7], p[8]); CSWAP(p[0], p[1]); CSWAP(p[3], p[4]); CSWAP(p[6], p[7]);
    CSWAP(p[1], p[2]); CSWAP(p[4], p[5]); CSWAP(p[7], p[8]); CSWAP(p[0], p[3]); CSWAP(p[5], p[8]); CSWAP(p[4], p[7]);
    CSWAP(p[3], p[6]); CSWAP(p[1], p[4]); CSWAP(p[2], p[5]); CSWAP(p[4], p[7]); CSWAP(p[4], p[2]); CSWAP(p[6], p[4]);
    CSWAP(p[4], p[2]);
    return p[4];
}

/* median of 25 by a fixed compare-exchange network (the classic 99-exchange selection network) */
static inline float median25(float *p) {
    static const unsigned char net[][2] = {
        {0, 1},   {3, 4},   {2, 4},   {2, 3},   {6, 7},   {5, 7},   {5, 6},   {9, 10},  {8, 10},  {8, 9},   {12, 13},
        {11, 13}, {11, 12}, {15, 16}, {14, 16}, {14, 15}, {18, 19}, {17, 19}, {17, 18}, {21, 22}, {20, 22}, {20, 21},
        {23, 24}, {2, 5},   {3, 6},   {0, 6},   {0, 3},   {4, 7},   {1, 7},   {1, 4},   {11, 14}, {8, 14},  {8, 11},
        {12, 15}, {9, 15},  {9, 12},  {13, 16}, {10, 16}, {10, 13}, {20, 23}, {17, 23}, {17, 20}, {21, 24}, {18, 24},
        {18, 21}, {19, 22}, {8, 17},  {9, 18},  {0, 18},  {0, 9},   {10, 19}, {1, 19},  {1, 10},  {11, 20}, {2, 20},
        {2, 11},  {12, 21}, {3, 21},  {3, 12},  {13, 22}, {4, 22},  {4, 13},  {14, 23}, {5, 23},  {5, 14},  {15, 24},
        {6, 24},  {6, 15},  {7, 16},  {7, 19},  {13, 21}, {15, 23}, {7, 13},  {7, 15},  {1, 9},   {3, 11},  {5, 17},
        {11, 17}, {9, 17},  {4, 10},  {6, 12},  {7, 14},  {4, 6},   {4, 7},   {12, 14}, {10, 14}, {6, 7},   {10, 12},
        {6, 10},  {6, 17},  {12, 17}, {7, 17},  {7, 10},  {12, 18}, {7, 12},  {10, 18}, {12, 20}, {10, 20}, {10, 12}};
    for (unsigned i = 0; i < sizeof net / sizeof net[0]; ++i)
        CSWAP(p[net[i][0]], p[net[i][1]]);
    return p[12];
}

void cpu_tvl1_median_blur(const float *src, float *dst, int w, int h, int ksize) {
    const int r = ksize / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *rows[5];
        for (int j = -r; j <= r; ++j) {
            int yy = y + j;
            yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
            rows[j + r] = src + (size_t)yy * w;
        }
        for (int x = 0; x < w; ++x) {
            float p[25];
            int n = 0;
            for (int j = 0; j < ksize; ++j)
                for (int i = -r; i <= r; ++i) {
                    int xx = x + i;
                    xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                    p[n++] = rows[j][xx];
                }
            dst[(size_t)y * w + x] = (ksize == 3) ? median9(p) : median25(p);
        }
    }
}

/* ---- cv::remap(INTER_CUBIC), CV_32F source, CV_32F maps, BORDER_CONSTANT(0) ----
 * imgwarp.cpp: coordinates are rounded to 1/32 px (INTER_BITS = 5), the 4x4 weights come from a 32x32 table
 * built with interpolateCubic (A = -0.75); taps outside the image contribute the border value 0. */
void cpu_tvl1_cubic_coeffs(float x, float *c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static float g_cubic_tab[32 * 32 * 16];
static int g_cubic_tab_ready = 0;

static void build_cubic_tab(void) {
    float t1[32][4];
    for (int i = 0; i < 32; ++i)
        cpu_tvl1_cubic_coeffs((float)i * (1.f / 32.f), t1[i]);
    for (int iy = 0; iy < 32; ++iy)
        for (int ix = 0; ix < 32; ++ix) {
            float *t = g_cubic_tab + (iy * 32 + ix) * 16;
            for (int k1 = 0; k1 < 4; ++k1)
                for (int k2 = 0; k2 < 4; ++k2)
                    t[k1 * 4 + k2] = t1[iy][k1] * t1[ix][k2];
        }
    g_cubic_tab_ready = 1;
}

void cpu_tvl1_remap_cubic(const float *src, int w, int h, const float *mapx, const float *mapy, float *dst) {
    if (!g_cubic_tab_ready) {
#pragma omp critical(cpu_tvl1_tab)
        if (!g_cubic_tab_ready)
            build_cubic_tab();
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const size_t o = (size_t)y * w + x;
            /* saturating like the short XY map upstream; keeps wild flows finite */
            double fxm = (double)mapx[o] * 32.0, fym = (double)mapy[o] * 32.0;
            if (!(fxm > -1048576.0))
                fxm = -1048576.0;
            if (fxm > 1048576.0)
                fxm = 1048576.0;
            if (!(fym > -1048576.0))
                fym = -1048576.0;
            if (fym > 1048576.0)
                fym = 1048576.0;
            const int sxq = (int)lrint(fxm), syq = (int)lrint(fym);
            const int sx = (sxq >> 5) - 1, sy = (syq >> 5) - 1;
            const float *wt = g_cubic_tab + (((syq & 31) * 32) + (sxq & 31)) * 16;
            float sum = 0.f;
            if (sx >= 0 && sy >= 0 && sx + 3 < w && sy + 3 < h) {
                const float *S = src + (size_t)sy * w + sx;
                for (int k1 = 0; k1 < 4; ++k1, S += w)
                    sum += S[0] * wt[k1 * 4] + S[1] * wt[k1 * 4 + 1] + S[2] * wt[k1 * 4 + 2] + S[3] * wt[k1 * 4 + 3];
            } else if (sx + 3 >= 0 && sy + 3 >= 0 && sx < w && sy < h) {
                for (int k1 = 0; k1 < 4; ++k1) {
                    const int yy = sy + k1;
                    if (yy < 0 || yy >= h)
                        continue; /* border value 0 */
                    for (int k2 = 0; k2 < 4; ++k2) {
                        const int xx = sx + k2;
                        if (xx >= 0 && xx < w)
                            sum += src[(size_t)yy * w + xx] * wt[k1 * 4 + k2];
                    }
                }
            }
            dst[o] = sum;
        }
    }
}

/* ---- the per-iteration passes of tvl1flow.cpp, one function per upstream *Body ---- */

static void centered_gradient(const float *src, float *dx, float *dy, int w, int h) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
        const float *r = src + (size_t)y * w, *rm = src + (size_t)ym * w, *rp = src + (size_t)yp * w;
        for (int x = 0; x < w; ++x) {
            const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
            dx[(size_t)y * w + x] = 0.5f * (r[xp] - r[xm]);
            dy[(size_t)y * w + x] = 0.5f * (rp[x] - rm[x]);
        }
    }
}

static void forward_gradient(const float *src, float *dx, float *dy, int w, int h) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *r = src + (size_t)y * w, *rp = src + (size_t)(y < h - 1 ? y + 1 : y) * w;
        float *X = dx + (size_t)y * w, *Y = dy + (size_t)y * w;
        for (int x = 0; x < w - 1; ++x) {
            X[x] = r[x + 1] - r[x];
            Y[x] = rp[x] - r[x];
        }
        X[w - 1] = 0.f;
        Y[w - 1] = rp[w - 1] - r[w - 1];
    }
}

static void divergence(const float *v1, const float *v2, float *div, int w, int h) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *a = v1 + (size_t)y * w, *b = v2 + (size_t)y * w;
        float *d = div + (size_t)y * w;
        if (y == 0) {
            d[0] = a[0] + b[0];
            for (int x = 1; x < w; ++x)
                d[x] = a[x] - a[x - 1] + b[x];
        } else {
            const float *bu = v2 + (size_t)(y - 1) * w;
            d[0] = a[0] + b[0] - bu[0];
            for (int x = 1; x < w; ++x)
                d[x] = (a[x] - a[x - 1]) + (b[x] - bu[x]);
        }
    }
}

static void calc_grad_rho(const float *I0, const float *I1w, const float *I1wx, const float *I1wy, const float *u1,
                          const float *u2, float *grad, float *rho_c, size_t n) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        const float Ix2 = I1wx[i] * I1wx[i], Iy2 = I1wy[i] * I1wy[i];
        grad[i] = Ix2 + Iy2;
        rho_c[i] = (I1w[i] - I1wx[i] * u1[i] - I1wy[i] * u2[i] - I0[i]);
    }
}

static void estimate_v(const float *I1wx, const float *I1wy, const float *u1, const float *u2, const float *grad,
                       const float *rho_c, float *v1, float *v2, float l_t, size_t n) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        const float rho = rho_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
        float d1 = 0.f, d2 = 0.f;
        if (rho < -l_t * grad[i]) {
            d1 = l_t * I1wx[i];
            d2 = l_t * I1wy[i];
        } else if (rho > l_t * grad[i]) {
            d1 = -l_t * I1wx[i];
            d2 = -l_t * I1wy[i];
        } else if (grad[i] > FLT_EPSILON) {
            const float fi = -rho / grad[i];
            d1 = fi * I1wx[i];
            d2 = fi * I1wy[i];
        }
        v1[i] = u1[i] + d1;
        v2[i] = u2[i] + d2;
    }
}

static float estimate_u(const float *v1, const float *v2, const float *div_p1, const float *div_p2, float *u1,
                        float *u2, float theta, size_t n) {
    double error = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : error)
    for (long long i = 0; i < (long long)n; ++i) {
        const float u1o = u1[i], u2o = u2[i];
        const float a = v1[i] + theta * div_p1[i], b = v2[i] + theta * div_p2[i];
        u1[i] = a;
        u2[i] = b;
        error += (double)((a - u1o) * (a - u1o) + (b - u2o) * (b - u2o));
    }
    return (float)error;
}

static void estimate_dual(const float *u1x, const float *u1y, const float *u2x, const float *u2y, float *p11,
                          float *p12, float *p21, float *p22, float taut, size_t n) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        const float g1 = hypotf(u1x[i], u1y[i]), g2 = hypotf(u2x[i], u2y[i]);
        const float ng1 = 1.f + taut * g1, ng2 = 1.f + taut * g2;
        p11[i] = (p11[i] + taut * u1x[i]) / ng1;
        p12[i] = (p12[i] + taut * u1y[i]) / ng1;
        p21[i] = (p21[i] + taut * u2x[i]) / ng2;
        p22[i] = (p22[i] + taut * u2y[i]) / ng2;
    }
}

typedef struct {
    float *I1x, *I1y, *map1, *map2, *I1w, *I1wx, *I1wy, *grad, *rho_c, *v1, *v2, *p11, *p12, *p21, *p22, *div_p1,
        *div_p2, *u1x, *u1y, *u2x, *u2y, *tmp;
} work_t;

static void proc_one_scale(const float *I0, const float *I1, float *u1, float *u2, int w, int h,
                           const cpu_tvl1_params *P, work_t *K, cpu_tvl1_stats *st) {
    const size_t n = (size_t)w * h;
    const float scaled_eps = (float)(P->epsilon * P->epsilon * (double)n);
    const float l_t = (float)(P->lambda * P->theta), taut = (float)(P->tau / P->theta), theta = (float)P->theta;
    centered_gradient(I1, K->I1x, K->I1y, w, h);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i)
        K->p11[i] = K->p12[i] = K->p21[i] = K->p22[i] = 0.f;

    for (int warpings = 0; warpings < P->warps; ++warpings) {
        /* buildFlowMap + three remaps + calcGradRho */
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t o = (size_t)y * w + x;
                K->map1[o] = (float)x + u1[o];
                K->map2[o] = (float)y + u2[o];
            }
        cpu_tvl1_remap_cubic(I1, w, h, K->map1, K->map2, K->I1w);
        cpu_tvl1_remap_cubic(K->I1x, w, h, K->map1, K->map2, K->I1wx);
        cpu_tvl1_remap_cubic(K->I1y, w, h, K->map1, K->map2, K->I1wy);
        calc_grad_rho(I0, K->I1w, K->I1wx, K->I1wy, u1, u2, K->grad, K->rho_c, n);

        float error = FLT_MAX;
        for (int n_outer = 0; error > scaled_eps && n_outer < P->outer_iterations; ++n_outer) {
            if (P->median_filtering > 1) {
                cpu_tvl1_median_blur(u1, K->tmp, w, h, P->median_filtering);
                memcpy(u1, K->tmp, n * sizeof(float));
                cpu_tvl1_median_blur(u2, K->tmp, w, h, P->median_filtering);
                memcpy(u2, K->tmp, n * sizeof(float));
            }
            if (st)
                st->outer_iterations += 1;
            for (int n_inner = 0; error > scaled_eps && n_inner < P->inner_iterations; ++n_inner) {
                estimate_v(K->I1wx, K->I1wy, u1, u2, K->grad, K->rho_c, K->v1, K->v2, l_t, n);
                divergence(K->p11, K->p12, K->div_p1, w, h);
                divergence(K->p21, K->p22, K->div_p2, w, h);
                error = estimate_u(K->v1, K->v2, K->div_p1, K->div_p2, u1, u2, theta, n);
                forward_gradient(u1, K->u1x, K->u1y, w, h);
                forward_gradient(u2, K->u2x, K->u2y, w, h);
                estimate_dual(K->u1x, K->u1y, K->u2x, K->u2y, K->p11, K->p12, K->p21, K->p22, taut, n);
                if (st) {
                    st->inner_iterations += 1;
                    st->px_iterations += (double)n;
                }
            }
        }
    }
}

int cpu_tvl1_calc(const uint8_t *I0, size_t pitch0, const uint8_t *I1, size_t pitch1, int W, int H,
                  const cpu_tvl1_params *params, float *flow_uv, cpu_tvl1_stats *stats) {
    cpu_tvl1_params P;
    if (params)
        P = *params;
    else
        cpu_tvl1_default_params(&P);
    if (W < 1 || H < 1 || P.nscales < 1 || P.nscales > CPU_TVL1_MAX_SCALES)
        return -1;
    if (stats)
        memset(stats, 0, sizeof *stats);
    const size_t n0 = (size_t)W * H;
    float *I0s[CPU_TVL1_MAX_SCALES] = {0}, *I1s[CPU_TVL1_MAX_SCALES] = {0}, *u1s[CPU_TVL1_MAX_SCALES] = {0},
          *u2s[CPU_TVL1_MAX_SCALES] = {0};
    int ws[CPU_TVL1_MAX_SCALES], hs[CPU_TVL1_MAX_SCALES];
    int nscales = P.nscales;
    ws[0] = W;
    hs[0] = H;
    I0s[0] = (float *)malloc(n0 * sizeof(float));
    I1s[0] = (float *)malloc(n0 * sizeof(float));
    u1s[0] = (float *)malloc(n0 * sizeof(float));
    u2s[0] = (float *)malloc(n0 * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            I0s[0][(size_t)y * W + x] = (float)I0[(size_t)y * pitch0 + x];
            I1s[0][(size_t)y * W + x] = (float)I1[(size_t)y * pitch1 + x];
        }
    for (int s = 1; s < P.nscales; ++s) {
        ws[s] = orc_cvround(ws[s - 1] * P.scale_step);
        hs[s] = orc_cvround(hs[s - 1] * P.scale_step);
        if (ws[s] < 1 || hs[s] < 1) {
            nscales = s;
            break;
        }
        const size_t n = (size_t)ws[s] * hs[s];
        I0s[s] = (float *)malloc(n * sizeof(float));
        I1s[s] = (float *)malloc(n * sizeof(float));
        cpu_tvl1_resize_linear(I0s[s - 1], ws[s - 1], hs[s - 1], I0s[s], ws[s], hs[s], P.scale_step, P.scale_step);
        cpu_tvl1_resize_linear(I1s[s - 1], ws[s - 1], hs[s - 1], I1s[s], ws[s], hs[s], P.scale_step, P.scale_step);
        if (ws[s] < 16 || hs[s] < 16) {
            nscales = s;
            break;
        }
        u1s[s] = (float *)malloc(n * sizeof(float));
        u2s[s] = (float *)malloc(n * sizeof(float));
    }
    work_t K;
    float **kp = (float **)&K;
    const int nbuf = (int)(sizeof K / sizeof(float *));
    for (int i = 0; i < nbuf; ++i) {
        kp[i] = (float *)malloc(n0 * sizeof(float));
        float *b = kp[i];
#pragma omp parallel for schedule(static) /* first touch by the threads that will use the rows */
        for (long long j = 0; j < (long long)n0; ++j)
            b[j] = 0.f;
    }
    {
        const size_t n = (size_t)ws[nscales - 1] * hs[nscales - 1];
        memset(u1s[nscales - 1], 0, n * sizeof(float));
        memset(u2s[nscales - 1], 0, n * sizeof(float));
    }
    for (int s = nscales - 1; s >= 0; --s) {
        proc_one_scale(I0s[s], I1s[s], u1s[s], u2s[s], ws[s], hs[s], &P, &K, stats);
        if (s == 0)
            break;
        cpu_tvl1_resize_linear(u1s[s], ws[s], hs[s], u1s[s - 1], ws[s - 1], hs[s - 1], (double)ws[s - 1] / ws[s],
                               (double)hs[s - 1] / hs[s]);
        cpu_tvl1_resize_linear(u2s[s], ws[s], hs[s], u2s[s - 1], ws[s - 1], hs[s - 1], (double)ws[s - 1] / ws[s],
                               (double)hs[s - 1] / hs[s]);
        const float up = (float)(1.0 / P.scale_step);
        const size_t n = (size_t)ws[s - 1] * hs[s - 1];
        float *a = u1s[s - 1], *b = u2s[s - 1];
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)n; ++i) {
            a[i] *= up;
            b[i] *= up;
        }
    }
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n0; ++i) {
        flow_uv[2 * i] = u1s[0][i];
        flow_uv[2 * i + 1] = u2s[0][i];
    }
    if (stats) {
        stats->nscales = nscales;
        for (int s = 0; s < nscales; ++s) {
            stats->w[s] = ws[s];
            stats->h[s] = hs[s];
        }
    }
    for (int i = 0; i < nbuf; ++i)
        free(kp[i]);
    for (int s = 0; s < CPU_TVL1_MAX_SCALES; ++s) {
        free(I0s[s]);
        free(I1s[s]);
        free(u1s[s]);
        free(u2s[s]);
    }
    return 0;
}
