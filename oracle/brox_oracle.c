/*
 * oracle/brox_oracle.c — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (spec confidence LOW).
 * Implements exactly the definition written in brox_oracle.h (which follows the structure of
 * cv::cuda::BroxOpticalFlow / NCVBroxOpticalFlow as used at /root/reference/src/denseflow_gpu.cpp:303,
 * :332-334).  float32, no FMA contraction, explicit operation order.
 */
#include "brox_oracle.h"

#include <float.h>

#define BROX_MAX_LEVELS 128
#define EPS2 1e-6f

void orc_brox_default_params(orc_brox_params *p) {
    p->alpha = 0.197f;
    p->gamma = 50.0f;
    p->scale_factor = 0.8f;
    p->inner_iterations = 10;
    p->outer_iterations = 77;
    p->solver_iterations = 10;
}

int orc_brox_pyramid_sizes(int W, int H, const orc_brox_params *p, int *wh, int max_levels) {
    int n = 0;
    float scale = 1.0f;
    int pw = W, ph = H;
    wh[0] = W;
    wh[1] = H;
    n = 1;
    while (pw > 15 && ph > 15 && n < p->outer_iterations && n < max_levels) {
        scale *= p->scale_factor;
        const int w = (int)ceilf((float)W * scale), h = (int)ceilf((float)H * scale);
        wh[2 * n] = w;
        wh[2 * n + 1] = h;
        ++n;
        pw = w;
        ph = h;
    }
    return n;
}

/* ---- resampling */

void orc_brox_downsample(const float *src, int sw, int sh, float *dst, int dw, int dh, float factor) {
    const float s = 1.0f / factor;
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < dh; ++iy) {
        const float y = s * (float)iy;
        const int yb = (int)floorf(y), ye = (int)ceilf(y + s);
        for (int ix = 0; ix < dw; ++ix) {
            const float x = s * (float)ix;
            const int xb = (int)floorf(x), xe = (int)ceilf(x + s);
            float sum = 0.f, wsum = 0.f;
            for (int cy = yb; cy < ye; ++cy) {
                const float wy = fminf((float)cy + 1.0f, y + s) - fmaxf((float)cy, y);
                const float *row = src + (size_t)orc_imin(cy, sh - 1) * sw;
                for (int cx = xb; cx < xe; ++cx) {
                    const float wx = fminf((float)cx + 1.0f, x + s) - fmaxf((float)cx, x);
                    const float w = wx * wy;
                    const float t = w * row[orc_imin(cx, sw - 1)];
                    sum = sum + t;
                    wsum = wsum + w;
                }
            }
            dst[(size_t)iy * dw + ix] = sum / wsum;
        }
    }
}

static inline float bicubic_w(float x_) {
    const float x = fabsf(x_);
    if (x <= 1.0f)
        return x * x * (1.5f * x - 2.5f) + 1.0f;
    else if (x < 2.0f)
        return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

void orc_brox_upsample_bicubic(const float *src, int sw, int sh, float *dst, int dw, int dh, float factor, float mul) {
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < dh; ++iy) {
        const float y = (float)iy * factor;
        const int y0 = orc_imax((int)ceilf(y - 2.0f), 0), y1 = orc_imin((int)floorf(y + 2.0f), sh - 1);
        for (int ix = 0; ix < dw; ++ix) {
            const float x = (float)ix * factor;
            const int x0 = orc_imax((int)ceilf(x - 2.0f), 0), x1 = orc_imin((int)floorf(x + 2.0f), sw - 1);
            float sum = 0.f, wsum = 0.f;
            for (int cy = y0; cy <= y1; ++cy) {
                const float wy = bicubic_w(y - (float)cy);
                for (int cx = x0; cx <= x1; ++cx) {
                    const float w = bicubic_w(x - (float)cx) * wy;
                    const float t = w * src[(size_t)cy * sw + cx];
                    sum = sum + t;
                    wsum = wsum + w;
                }
            }
            const float v = (wsum == 0.0f) ? 0.0f : sum / wsum;
            dst[(size_t)iy * dw + ix] = v * mul;
        }
    }
}

/* ---- derivatives: (1,-8,0,8,-1)/12, mirror border with edge duplication */

static inline int mirror_idx(int i, int n) {
    const int p = 2 * n;
    int m = i % p;
    if (m < 0)
        m += p;
    return m < n ? m : p - 1 - m;
}

static const float kD[5] = {1.0f, -8.0f, 0.0f, 8.0f, -1.0f};

void orc_brox_deriv_x(const float *src, int w, int h, float *dst) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = 0.f;
            for (int k = 0; k < 5; ++k) {
                const float t = src[(size_t)y * w + mirror_idx(x + k - 2, w)] * kD[k];
                s = s + t;
            }
            dst[(size_t)y * w + x] = s * (1.0f / 12.0f);
        }
}

void orc_brox_deriv_y(const float *src, int w, int h, float *dst) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = 0.f;
            for (int k = 0; k < 5; ++k) {
                const float t = src[(size_t)mirror_idx(y + k - 2, h) * w + x] * kD[k];
                s = s + t;
            }
            dst[(size_t)y * w + x] = s * (1.0f / 12.0f);
        }
}

/* ---- bilinear sampling with mirror indices at (fx, fy) in pixel coordinates */

typedef struct {
    int i00, i01, i10, i11;
    float ax, ay;
} bl_tap;

static inline bl_tap bl_setup(float fx, float fy, int w, int h) {
    bl_tap t;
    fx = fminf(fmaxf(fx, -1.0e6f), 1.0e6f); /* keeps the int conversion defined for wild flows */
    fy = fminf(fmaxf(fy, -1.0e6f), 1.0e6f);
    const float x0 = floorf(fx), y0 = floorf(fy);
    t.ax = fx - x0;
    t.ay = fy - y0;
    const int xa = mirror_idx((int)x0, w), xb = mirror_idx((int)x0 + 1, w);
    const int ya = mirror_idx((int)y0, h), yb = mirror_idx((int)y0 + 1, h);
    t.i00 = ya * w + xa;
    t.i01 = ya * w + xb;
    t.i10 = yb * w + xa;
    t.i11 = yb * w + xb;
    return t;
}

static inline float bl_sample(const float *p, const bl_tap *t) {
    const float a = (1.0f - t->ax) * p[t->i00] + t->ax * p[t->i01];
    const float b = (1.0f - t->ax) * p[t->i10] + t->ax * p[t->i11];
    return (1.0f - t->ay) * a + t->ay * b;
}

static inline float inv_sqrt(float s) { return 1.0f / sqrtf(s); }

/* ---- one pyramid level */

typedef struct {
    int w, h;
    const float *I0, *I1;
    float *Ix0, *Iy0, *Ix, *Iy, *Ixx, *Ixy, *Iyy;
    float *du, *dv, *gx, *gy, *inv_den_u, *inv_den_v, *num_dudv, *num_u, *num_v;
} brox_level;

static void stage1(const brox_level *L, const float *u, const float *v, float alpha, float gamma) {
    const int w = L->w, h = L->h;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const int ym = orc_imax(y - 1, 0), yp = orc_imin(y + 1, h - 1);
        for (int x = 0; x < w; ++x) {
            const int xm = orc_imax(x - 1, 0), xp = orc_imin(x + 1, w - 1);
            const size_t o = (size_t)y * w + x;
#define WU(xx, yy) (u[(size_t)(yy)*w + (xx)] + L->du[(size_t)(yy)*w + (xx)])
#define WV(xx, yy) (v[(size_t)(yy)*w + (xx)] + L->dv[(size_t)(yy)*w + (xx)])
            /* data term */
            const bl_tap t = bl_setup((float)x + u[o], (float)y + v[o], w, h);
            const float I1w = bl_sample(L->I1, &t), Ixw = bl_sample(L->Ix, &t), Iyw = bl_sample(L->Iy, &t);
            const float Ixxw = bl_sample(L->Ixx, &t), Ixyw = bl_sample(L->Ixy, &t), Iyyw = bl_sample(L->Iyy, &t);
            const float Iz = I1w - L->I0[o], Ixz = Ixw - L->Ix0[o], Iyz = Iyw - L->Iy0[o];
            const float du = L->du[o], dv = L->dv[o];
            const float q0 = Iz + (Ixw * du + Iyw * dv);
            const float q1 = Ixz + (Ixxw * du + Ixyw * dv);
            const float q2 = Iyz + (Ixyw * du + Iyyw * dv);
            const float psi = (0.5f * inv_sqrt((q0 * q0 + gamma * (q1 * q1 + q2 * q2)) + EPS2)) / alpha;
            L->num_dudv[o] = psi * (Ixw * Iyw + gamma * (Ixxw * Ixyw + Ixyw * Iyyw));
            L->inv_den_u[o] = psi * (Ixw * Ixw + gamma * (Ixyw * Ixyw + Ixxw * Ixxw)); /* den_u until stage 2 */
            L->inv_den_v[o] = psi * (Iyw * Iyw + gamma * (Ixyw * Ixyw + Iyyw * Iyyw));
            L->num_u[o] = psi * (Ixw * Iz + gamma * (Ixxw * Ixz + Ixyw * Iyz));
            L->num_v[o] = psi * (Iyw * Iz + gamma * (Iyyw * Iyz + Ixyw * Ixz));
            /* diffusivities on the staggered grid */
            if (x > 0) {
                const float ux = WU(x, y) - WU(xm, y), vx = WV(x, y) - WV(xm, y);
                const float uy = 0.25f * (((WU(x, yp) + WU(xm, yp)) - WU(x, ym)) - WU(xm, ym));
                const float vy = 0.25f * (((WV(x, yp) + WV(xm, yp)) - WV(x, ym)) - WV(xm, ym));
                L->gx[o] = 0.5f * inv_sqrt((((ux * ux + uy * uy) + vx * vx) + vy * vy) + EPS2);
            } else {
                L->gx[o] = 0.0f;
            }
            if (y > 0) {
                const float uy = WU(x, y) - WU(x, ym), vy = WV(x, y) - WV(x, ym);
                const float ux = 0.25f * (((WU(xp, y) + WU(xp, ym)) - WU(xm, y)) - WU(xm, ym));
                const float vx = 0.25f * (((WV(xp, y) + WV(xp, ym)) - WV(xm, y)) - WV(xm, ym));
                L->gy[o] = 0.5f * inv_sqrt((((ux * ux + uy * uy) + vx * vx) + vy * vy) + EPS2);
            } else {
                L->gy[o] = 0.0f;
            }
#undef WU
#undef WV
        }
    }
}

static void stage2(const brox_level *L) {
    const int w = L->w, h = L->h;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t o = (size_t)y * w + x;
            const float gl = L->gx[o], gr = (x + 1 < w) ? L->gx[o + 1] : 0.0f;
            const float gd = L->gy[o], gu = (y + 1 < h) ? L->gy[o + w] : 0.0f;
            const float gs = ((gl + gr) + gd) + gu;
            L->inv_den_u[o] = 1.0f / (L->inv_den_u[o] + gs);
            L->inv_den_v[o] = 1.0f / (L->inv_den_v[o] + gs);
        }
}

static void sor_pass(const brox_level *L, const float *u, const float *v, int color, float omega) {
    const int w = L->w, h = L->h;
    const int jacobi = (orc_get_variant() & ORC_VAR_BROX_JACOBI) != 0;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = (y + color) & 1; x < w; x += 2) {
            const size_t o = (size_t)y * w + x;
            const int xm = orc_imax(x - 1, 0), xp = orc_imin(x + 1, w - 1);
            const int ym = orc_imax(y - 1, 0), yp = orc_imin(y + 1, h - 1);
            const size_t ol = (size_t)y * w + xm, orr = (size_t)y * w + xp, od = (size_t)ym * w + x,
                         ou = (size_t)yp * w + x;
            const float gl = L->gx[o], gr = (x + 1 < w) ? L->gx[o + 1] : 0.0f;
            const float gd = L->gy[o], gu = (y + 1 < h) ? L->gy[o + w] : 0.0f;
            const float gs = ((gl + gr) + gd) + gu;
            const float su = (((gl * (u[ol] + L->du[ol]) + gr * (u[orr] + L->du[orr])) + gd * (u[od] + L->du[od])) +
                              gu * (u[ou] + L->du[ou])) -
                             gs * u[o];
            const float sv = (((gl * (v[ol] + L->dv[ol]) + gr * (v[orr] + L->dv[orr])) + gd * (v[od] + L->dv[od])) +
                              gu * (v[ou] + L->dv[ou])) -
                             gs * v[o];
            const float du = L->du[o], dv = L->dv[o];
            const float du_n = (1.0f - omega) * du + omega * (L->inv_den_u[o] * ((su - L->num_u[o]) - L->num_dudv[o] * dv));
            const float du_c = jacobi ? du : du_n; /* upstream overwrites du before it forms dv (Gauss-Seidel) */
            const float dv_n = (1.0f - omega) * dv + omega * (L->inv_den_v[o] * ((sv - L->num_v[o]) - L->num_dudv[o] * du_c));
            L->du[o] = du_n;
            L->dv[o] = dv_n;
        }
}

int orc_brox_calc(const uint8_t *I0u8, size_t pitch0, const uint8_t *I1u8, size_t pitch1, int W, int H,
                  const orc_brox_params *params, float *flow_uv) {
    orc_brox_params prm;
    if (params)
        prm = *params;
    else
        orc_brox_default_params(&prm);
    if (W < 1 || H < 1 || !(prm.scale_factor > 0.f && prm.scale_factor < 1.f) || prm.inner_iterations < 0 ||
        prm.outer_iterations < 1 || prm.solver_iterations < 0 || !(prm.alpha > 0.f))
        return -1;
    int wh[2 * BROX_MAX_LEVELS];
    const int nl = orc_brox_pyramid_sizes(W, H, &prm, wh, BROX_MAX_LEVELS);
    float *P0[BROX_MAX_LEVELS], *P1[BROX_MAX_LEVELS];
    const size_t N0 = (size_t)W * H;
    P0[0] = (float *)malloc(sizeof(float) * N0);
    P1[0] = (float *)malloc(sizeof(float) * N0);
    const float a255 = (float)(1.0 / 255.0);
    orc_convert_u8_f32(I0u8, pitch0, W, H, a255, P0[0]);
    orc_convert_u8_f32(I1u8, pitch1, W, H, a255, P1[0]);
    for (int l = 1; l < nl; ++l) {
        const int w = wh[2 * l], h = wh[2 * l + 1];
        P0[l] = (float *)malloc(sizeof(float) * (size_t)w * h);
        P1[l] = (float *)malloc(sizeof(float) * (size_t)w * h);
        orc_brox_downsample(P0[l - 1], wh[2 * l - 2], wh[2 * l - 1], P0[l], w, h, prm.scale_factor);
        orc_brox_downsample(P1[l - 1], wh[2 * l - 2], wh[2 * l - 1], P1[l], w, h, prm.scale_factor);
    }
    float *buf = (float *)malloc(sizeof(float) * N0 * 20);
    float *u = buf, *v = buf + N0, *u2 = buf + 2 * N0, *v2 = buf + 3 * N0;
    brox_level L;
    L.Ix0 = buf + 4 * N0;
    L.Iy0 = buf + 5 * N0;
    L.Ix = buf + 6 * N0;
    L.Iy = buf + 7 * N0;
    L.Ixx = buf + 8 * N0;
    L.Ixy = buf + 9 * N0;
    L.Iyy = buf + 10 * N0;
    L.du = buf + 11 * N0;
    L.dv = buf + 12 * N0;
    L.gx = buf + 13 * N0;
    L.gy = buf + 14 * N0;
    L.inv_den_u = buf + 15 * N0;
    L.inv_den_v = buf + 16 * N0;
    L.num_dudv = buf + 17 * N0;
    L.num_u = buf + 18 * N0;
    L.num_v = buf + 19 * N0;
    {
        const size_t nc = (size_t)wh[2 * (nl - 1)] * wh[2 * (nl - 1) + 1];
        memset(u, 0, sizeof(float) * nc);
        memset(v, 0, sizeof(float) * nc);
    }
    const float omega = orc_get_brox_omega(); /* 1.99f unless a test changes it */
    for (int l = nl - 1; l >= 0; --l) {
        const int w = wh[2 * l], h = wh[2 * l + 1];
        const size_t n = (size_t)w * h;
        L.w = w;
        L.h = h;
        L.I0 = P0[l];
        L.I1 = P1[l];
        orc_brox_deriv_x(L.I0, w, h, L.Ix0);
        orc_brox_deriv_y(L.I0, w, h, L.Iy0);
        orc_brox_deriv_x(L.I1, w, h, L.Ix);
        orc_brox_deriv_y(L.I1, w, h, L.Iy);
        orc_brox_deriv_x(L.Ix, w, h, L.Ixx);
        orc_brox_deriv_y(L.Iy, w, h, L.Iyy);
        orc_brox_deriv_y(L.Ix, w, h, L.Ixy);
        memset(L.du, 0, sizeof(float) * n);
        memset(L.dv, 0, sizeof(float) * n);
        for (int in = 0; in < prm.inner_iterations; ++in) {
            stage1(&L, u, v, prm.alpha, prm.gamma);
            stage2(&L);
            for (int si = 0; si < prm.solver_iterations; ++si) {
                sor_pass(&L, u, v, 0, omega);
                sor_pass(&L, u, v, 1, omega);
            }
        }
        for (size_t i = 0; i < n; ++i) {
            u[i] = u[i] + L.du[i];
            v[i] = v[i] + L.dv[i];
        }
        if (l > 0) {
            const int nw = wh[2 * l - 2], nh = wh[2 * l - 1];
            const float mul = 1.0f / prm.scale_factor;
            orc_brox_upsample_bicubic(u, w, h, u2, nw, nh, prm.scale_factor, mul);
            orc_brox_upsample_bicubic(v, w, h, v2, nw, nh, prm.scale_factor, mul);
            float *t;
            t = u, u = u2, u2 = t;
            t = v, v = v2, v2 = t;
        }
    }
    for (size_t i = 0; i < N0; ++i) {
        flow_uv[2 * i] = u[i];
        flow_uv[2 * i + 1] = v[i];
    }
    for (int l = 0; l < nl; ++l) {
        free(P0[l]);
        free(P1[l]);
    }
    free(buf);
    return 0;
}
