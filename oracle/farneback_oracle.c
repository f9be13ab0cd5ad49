/*
 * oracle/farneback_oracle.c — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * CPU restatement of cv::cuda::FarnebackOpticalFlow with create() defaults (numLevels 5,
 * pyrScale 0.5, winSize 13, numIters 10, polyN 5, polySigma 1.1, flags 0 => box-filter update),
 * the algorithm the reference invokes at /root/reference/src/denseflow_gpu.cpp:301 / :329.
 * The arithmetic is third-party (opencv_contrib 4.5.2 cudaoptflow) and absent from
 * /root/reference; this file restates it as written down in SURVEY.md Appendix B (B.1-B.9).
 *
 * float32, no FMA contraction, accumulation orders as in the upstream kernels.
 * One documented simplification: B.6 evaluates the Gaussian taps with libm exp() in double instead
 * of OpenCV's softfloat exp (<= 1 ulp(double) apart before the cast to float).
 */
#include "farneback_oracle.h"

#include <float.h>
#include <stdio.h>

#define MIN_SIZE 32
#define BORDER_SIZE 5

void orc_farneback_default_params(orc_farneback_params *p) {
    p->num_levels = 5;
    p->pyr_scale = 0.5;
    p->fast_pyramids = 0;
    p->win_size = 13;
    p->num_iters = 10;
    p->poly_n = 5;
    p->poly_sigma = 1.1;
    p->flags = 0;
}

/* ---------------------------------------------------------------- B.3 */

static int invert_spd6(double A[6][6], double inv[6][6]) {
    /* Cholesky A = L L^T, then solve for the inverse column by column (DECOMP_CHOLESKY) */
    double L[6][6];
    memset(L, 0, sizeof L);
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = A[i][j];
            for (int k = 0; k < j; ++k)
                s -= L[i][k] * L[j][k];
            if (i == j) {
                if (s <= 0)
                    return -1;
                L[i][i] = sqrt(s);
            } else {
                L[i][j] = s / L[j][j];
            }
        }
    }
    for (int c = 0; c < 6; ++c) {
        double y[6], x[6];
        for (int i = 0; i < 6; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k)
                s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
        for (int i = 5; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < 6; ++k)
                s -= L[k][i] * x[k];
            x[i] = s / L[i][i];
        }
        for (int i = 0; i < 6; ++i)
            inv[i][c] = x[i];
    }
    return 0;
}

void orc_farneback_prepare_poly(int n, double sigma, orc_farneback_poly_consts *out) {
    float gbuf[15 * 3];
    float *g = gbuf + n, *xg = g + n * 2 + 1, *xxg = xg + n * 2 + 1;
    if (sigma < FLT_EPSILON)
        sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6];
    memset(G, 0, sizeof G);
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            /* float products, exactly as the upstream expression evaluates them */
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    double inv[6][6];
    invert_spd6(G, inv);
    for (int i = 0; i <= n; ++i) {
        out->g[i] = g[i];
        out->xg[i] = xg[i];
        out->xxg[i] = xxg[i];
    }
    out->ig11 = (float)inv[1][1];
    out->ig03 = (float)inv[0][3];
    out->ig33 = (float)inv[3][3];
    out->ig55 = (float)inv[5][5];
}

/* ---------------------------------------------------------------- B.6 */

int orc_farneback_gaussian_kernel(int n, double sigma, float *k) {
    if (n < 1 || !(n & 1))
        return -1;
    if (sigma <= 0 && !(orc_get_variant() & ORC_VAR_FARN_SIGMA0_COMPUTED)) {
        static const double t1[] = {1.};
        static const double t3[] = {0.25, 0.5, 0.25};
        static const double t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
        static const double t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
        const double *t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : n == 7 ? t7 : NULL;
        if (t) {
            for (int i = 0; i < n; ++i)
                k[i] = (float)t[i];
            return 0;
        }
    }
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.125 / (sigmaX * sigmaX); /* taps are indexed by x = 2*(i - (n-1)/2) */
    const int n2 = (n - 1) / 2;
    double v[64];
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        v[i] = exp((double)(x * x) * scale2X);
        sum += v[i];
    }
    sum *= 2;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; ++i) {
        k[i] = (float)(v[i] * mul1);
        k[n - 1 - i] = k[i];
    }
    k[n2] = (float)mul1;
    return 0;
}

/* ---------------------------------------------------------------- B.4 */

static inline int reflect101(int x, int last) { /* BrdReflect101: idx_low(idx_high(x)) */
    int hi = abs(last - abs(last - x)) % (last + 1);
    return abs(hi) % (last + 1);
}
static inline int reflect101_low(int x, int last) { return abs(x) % (last + 1); }
static inline int reflect101_high(int x, int last) { return abs(last - abs(last - x)) % (last + 1); }

void orc_farneback_gaussian_blur(const float *src, int W, int H, const float *ker, int half, float *dst) {
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * (size_t)(W + 2 * half));
#pragma omp for schedule(static)
        for (int y = 0; y < H; ++y) {
            /* vertical pass into an extended row */
            for (int i = 0; i < W + 2 * half; ++i) {
                const int xe = reflect101(i - half, W - 1);
                float r = src[(size_t)y * W + xe] * ker[0];
                for (int j = 1; j <= half; ++j) {
                    const float a = src[(size_t)reflect101_low(y - j, H - 1) * W + xe];
                    const float b = src[(size_t)reflect101_high(y + j, H - 1) * W + xe];
                    const float t = (a + b) * ker[j];
                    r = r + t;
                }
                row[i] = r;
            }
            /* horizontal pass */
            for (int x = 0; x < W; ++x) {
                const float *c = row + x + half;
                float res = c[0] * ker[0];
                for (int i = 1; i <= half; ++i) {
                    const float t = (c[-i] + c[i]) * ker[i];
                    res = res + t;
                }
                dst[(size_t)y * W + x] = res;
            }
        }
        free(row);
    }
}

/* ---------------------------------------------------------------- B.5 */

void orc_farneback_poly_exp(const float *src, int W, int H, int n, const orc_farneback_poly_consts *c, float *R) {
    const size_t plane = (size_t)W * H;
#pragma omp parallel
    {
        float *r0 = (float *)malloc(sizeof(float) * (size_t)(W + 2 * n) * 3);
        float *r1 = r0 + (W + 2 * n), *r2 = r1 + (W + 2 * n);
#pragma omp for schedule(static)
        for (int y = 0; y < H; ++y) {
            for (int i = 0; i < W + 2 * n; ++i) {
                const int xw = orc_imin(orc_imax(i - n, 0), W - 1);
                float a0 = src[(size_t)y * W + xw] * c->g[0];
                float a1 = 0.f, a2 = 0.f;
                for (int k = 1; k <= n; ++k) {
                    const float t0 = src[(size_t)orc_imax(y - k, 0) * W + xw];
                    const float t1 = src[(size_t)orc_imin(y + k, H - 1) * W + xw];
                    float t;
                    t = c->g[k] * (t0 + t1);
                    a0 = a0 + t;
                    t = c->xg[k] * (t1 - t0);
                    a1 = a1 + t;
                    t = c->xxg[k] * (t0 + t1);
                    a2 = a2 + t;
                }
                r0[i] = a0;
                r1[i] = a1;
                r2[i] = a2;
            }
            for (int x = 0; x < W; ++x) {
                const float *p0 = r0 + x + n, *p1 = r1 + x + n, *p2 = r2 + x + n;
                float b1 = c->g[0] * p0[0];
                float b3 = c->g[0] * p1[0];
                float b5 = c->g[0] * p2[0];
                float b2 = 0, b4 = 0, b6 = 0;
                for (int k = 1; k <= n; ++k) {
                    float t;
                    t = (p0[k] + p0[-k]) * c->g[k];
                    b1 = b1 + t;
                    t = (p0[k] + p0[-k]) * c->xxg[k];
                    b4 = b4 + t;
                    t = (p0[k] - p0[-k]) * c->xg[k];
                    b2 = b2 + t;
                    t = (p1[k] + p1[-k]) * c->g[k];
                    b3 = b3 + t;
                    t = (p1[k] - p1[-k]) * c->xg[k];
                    b6 = b6 + t;
                    t = (p2[k] + p2[-k]) * c->g[k];
                    b5 = b5 + t;
                }
                const size_t o = (size_t)y * W + x;
                R[o] = b3 * c->ig11;
                R[plane + o] = b2 * c->ig11;
                {
                    const float u = b1 * c->ig03, v = b5 * c->ig33;
                    R[2 * plane + o] = u + v;
                }
                {
                    const float u = b1 * c->ig03, v = b4 * c->ig33;
                    R[3 * plane + o] = u + v;
                }
                R[4 * plane + o] = b6 * c->ig55;
            }
        }
        free(r0);
    }
}

/* ---------------------------------------------------------------- B.7 */

static const float c_border[BORDER_SIZE + 1] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f, 1.f};

void orc_farneback_update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1, int W,
                                   int H, float *M) {
    const size_t plane = (size_t)W * H;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const size_t o = (size_t)y * W + x;
            const float dx = flowx[o], dy = flowy[o];
            float fx = (float)x + dx;
            float fy = (float)y + dy;
            const int x1 = (int)floorf(fx);
            const int y1 = (int)floorf(fy);
            fx -= (float)x1;
            fy -= (float)y1;
            float r2, r3, r4, r5, r6;
            if (x1 >= 0 && y1 >= 0 && x1 < W - 1 && y1 < H - 1) {
                const float a00 = (1.f - fx) * (1.f - fy);
                const float a01 = fx * (1.f - fy);
                const float a10 = (1.f - fx) * fy;
                const float a11 = fx * fy;
                const size_t q = (size_t)y1 * W + x1;
                float v[5];
                for (int p = 0; p < 5; ++p) {
                    const float *Rp = R1 + p * plane;
                    float s = a00 * Rp[q];
                    float t;
                    t = a01 * Rp[q + 1];
                    s = s + t;
                    t = a10 * Rp[q + W];
                    s = s + t;
                    t = a11 * Rp[q + W + 1];
                    s = s + t;
                    v[p] = s;
                }
                r2 = v[0];
                r3 = v[1];
                r4 = (R0[2 * plane + o] + v[2]) * 0.5f;
                r5 = (R0[3 * plane + o] + v[3]) * 0.5f;
                r6 = (R0[4 * plane + o] + v[4]) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[2 * plane + o];
                r5 = R0[3 * plane + o];
                r6 = R0[4 * plane + o] * 0.5f;
            }
            r2 = (R0[o] - r2) * 0.5f;
            r3 = (R0[plane + o] - r3) * 0.5f;
            {
                const float a = r4 * dy, b = r6 * dx;
                r2 = r2 + (a + b); /* r2 += r4*dy + r6*dx : the right-hand side is summed first */
            }
            {
                const float a = r6 * dy, b = r5 * dx;
                r3 = r3 + (a + b);
            }
            float scale = c_border[orc_imin(x, BORDER_SIZE)] * c_border[orc_imin(y, BORDER_SIZE)];
            scale = scale * c_border[orc_imin(W - x - 1, BORDER_SIZE)];
            scale = scale * c_border[orc_imin(H - y - 1, BORDER_SIZE)];
            r2 *= scale;
            r3 *= scale;
            r4 *= scale;
            r5 *= scale;
            r6 *= scale;
            {
                const float a = r4 * r4, b = r6 * r6;
                M[o] = a + b;
            }
            M[plane + o] = (r4 + r5) * r6;
            {
                const float a = r5 * r5, b = r6 * r6;
                M[2 * plane + o] = a + b;
            }
            {
                const float a = r4 * r2, b = r6 * r3;
                M[3 * plane + o] = a + b;
            }
            {
                const float a = r6 * r2, b = r5 * r3;
                M[4 * plane + o] = a + b;
            }
        }
    }
}

/* ---------------------------------------------------------------- B.8 */

void orc_farneback_box_filter5(const float *src, int W, int H, int half, float *dst) {
    const size_t plane = (size_t)W * H;
    const float inv = 1.f / (float)((1 + 2 * half) * (1 + 2 * half));
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * (size_t)(W + 2 * half));
#pragma omp for schedule(static) collapse(2)
        for (int p = 0; p < 5; ++p) {
            for (int y = 0; y < H; ++y) {
                const float *S = src + p * plane;
                for (int i = 0; i < W + 2 * half; ++i) {
                    const int xe = orc_imin(orc_imax(i - half, 0), W - 1);
                    float r = S[(size_t)y * W + xe];
                    for (int j = 1; j <= half; ++j) {
                        const float t = S[(size_t)orc_imax(y - j, 0) * W + xe] + S[(size_t)orc_imin(y + j, H - 1) * W + xe];
                        r = r + t;
                    }
                    row[i] = r;
                }
                for (int x = 0; x < W; ++x) {
                    const float *c = row + x + half;
                    float res = c[0];
                    for (int i = 1; i <= half; ++i) {
                        const float t = c[-i] + c[i];
                        res = res + t;
                    }
                    dst[p * plane + (size_t)y * W + x] = res * inv;
                }
            }
        }
        free(row);
    }
}

/* ---------------------------------------------------------------- B.9 */

void orc_farneback_update_flow(const float *M, int W, int H, float *flowx, float *flowy) {
    const size_t plane = (size_t)W * H;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)plane; ++i) {
        const float g11 = M[i], g12 = M[plane + i], g22 = M[2 * plane + i], h1 = M[3 * plane + i], h2 = M[4 * plane + i];
        float det;
        {
            const float a = g11 * g22, b = g12 * g12;
            det = (a - b) + 1e-3f;
        }
        const float detInv = 1.f / det;
        {
            const float a = g11 * h2, b = g12 * h1;
            flowx[i] = (a - b) * detInv;
        }
        {
            const float a = g22 * h1, b = g12 * h2;
            flowy[i] = (a - b) * detInv;
        }
    }
}

/* ---------------------------------------------------------------- B.2 driver */

int orc_farneback_calc(const uint8_t *I0u8, size_t pitch0, const uint8_t *I1u8, size_t pitch1, int W, int H,
                       const orc_farneback_params *params, float *flow_uv) {
    orc_farneback_params prm;
    if (params)
        prm = *params;
    else
        orc_farneback_default_params(&prm);
    if ((prm.poly_n != 5 && prm.poly_n != 7) || prm.fast_pyramids || prm.flags != 0 || prm.win_size < 1 ||
        !(prm.win_size & 1) || prm.num_levels < 0 || !(prm.pyr_scale > 0 && prm.pyr_scale < 1) || W < 1 || H < 1)
        return -1;

    const size_t N0 = (size_t)W * H;
    float *frames[2];
    frames[0] = (float *)malloc(sizeof(float) * N0);
    frames[1] = (float *)malloc(sizeof(float) * N0);
    orc_convert_u8_f32(I0u8, pitch0, W, H, 1.0f, frames[0]);
    orc_convert_u8_f32(I1u8, pitch1, W, H, 1.0f, frames[1]);

    double scale = 1;
    int numLevelsCropped = 0;
    for (; numLevelsCropped < prm.num_levels; numLevelsCropped++) {
        scale *= prm.pyr_scale;
        if (W * scale < MIN_SIZE || H * scale < MIN_SIZE)
            break;
    }

    orc_farneback_poly_consts pc;
    orc_farneback_prepare_poly(prm.poly_n, prm.poly_sigma, &pc);

    float *blurred = (float *)malloc(sizeof(float) * N0);
    float *pyr = (float *)malloc(sizeof(float) * N0);
    float *R[2] = {(float *)malloc(sizeof(float) * N0 * 5), (float *)malloc(sizeof(float) * N0 * 5)};
    float *M = (float *)malloc(sizeof(float) * N0 * 5), *bufM = (float *)malloc(sizeof(float) * N0 * 5);
    float *curx = (float *)malloc(sizeof(float) * N0), *cury = (float *)malloc(sizeof(float) * N0);
    float *prevx = (float *)malloc(sizeof(float) * N0), *prevy = (float *)malloc(sizeof(float) * N0);
    int pw = 0, ph = 0;

    for (int k = numLevelsCropped; k >= 0; k--) {
        scale = 1;
        for (int i = 0; i < k; i++)
            scale *= prm.pyr_scale;
        const double sigma = (1. / scale - 1) * 0.5;
        int smoothSize = orc_cvround(sigma * 5) | 1;
        smoothSize = orc_imax(smoothSize, 3);
        const int width = orc_cvround(W * scale);
        const int height = orc_cvround(H * scale);
        const size_t n = (size_t)width * height;

        if (pw == 0) {
            memset(curx, 0, sizeof(float) * n);
            memset(cury, 0, sizeof(float) * n);
        } else {
            const float ifx = orc_inv_scale_from_sizes(width, pw), ify = orc_inv_scale_from_sizes(height, ph);
            orc_resize_linear(prevx, pw, ph, curx, width, height, ifx, ify);
            orc_resize_linear(prevy, pw, ph, cury, width, height, ifx, ify);
            const float up = (float)(1. / prm.pyr_scale);
            orc_mul_scalar(curx, n, up);
            orc_mul_scalar(cury, n, up);
        }

        float gk[512];
        if (smoothSize > 511 || orc_farneback_gaussian_kernel(smoothSize, sigma, gk) != 0)
            return -2;
        const int half = smoothSize / 2;
        for (int i = 0; i < 2; ++i) {
            orc_farneback_gaussian_blur(frames[i], W, H, gk + half, half, blurred);
            const float ifx = orc_inv_scale_from_sizes(width, W), ify = orc_inv_scale_from_sizes(height, H);
            orc_resize_linear(blurred, W, H, pyr, width, height, ifx, ify);
            orc_farneback_poly_exp(pyr, width, height, prm.poly_n, &pc, R[i]);
        }

        orc_farneback_update_matrices(curx, cury, R[0], R[1], width, height, M);
        for (int it = 0; it < prm.num_iters; ++it) {
            orc_farneback_box_filter5(M, width, height, prm.win_size / 2, bufM);
            float *t = M;
            M = bufM;
            bufM = t;
            orc_farneback_update_flow(M, width, height, curx, cury);
            if (it < prm.num_iters - 1)
                orc_farneback_update_matrices(curx, cury, R[0], R[1], width, height, M);
        }
        float *t;
        t = prevx, prevx = curx, curx = t;
        t = prevy, prevy = cury, cury = t;
        pw = width;
        ph = height;
    }
    /* level 0 has width == W */
    for (size_t i = 0; i < N0; ++i) {
        flow_uv[2 * i] = prevx[i];
        flow_uv[2 * i + 1] = prevy[i];
    }
    free(frames[0]); free(frames[1]); free(blurred); free(pyr); free(R[0]); free(R[1]);
    free(M); free(bufM); free(curx); free(cury); free(prevx); free(prevy);
    return 0;
}
