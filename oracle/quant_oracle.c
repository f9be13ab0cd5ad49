/*
 * oracle/quant_oracle.c — TEST INFRASTRUCTURE ONLY.  See quant_oracle.h.
 */
#include "quant_oracle.h"

#include "oracle_common.h"

/* src/common.cpp:6, the CAST macro: the float component is promoted to double, compared with the bounds,
 * and otherwise mapped by 255*(v-L)/(H-L) — multiply first, then divide — through cvRound (E.6: round
 * half to even; NaN fails both comparisons and cvRound(NaN) truncates to 0 in the 8-bit store). */
static uint8_t cast_bound(float vf, double lo, double hi) {
    const double v = (double)vf;
    if (v > hi)
        return 255;
    if (v < lo)
        return 0;
    return (uint8_t)orc_cvround(255 * (v - lo) / (hi - lo));
}

/* src/common.cpp:7-14: every pixel, x from the first plane and y from the second. */
void orc_flow_to_u8(const float *flow_uv, int w, int h, double lower_bound, double upper_bound, uint8_t *img_x,
                    uint8_t *img_y) {
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const size_t p = (size_t)i * w + j;
            img_x[p] = cast_bound(flow_uv[2 * p], lower_bound, upper_bound);
            img_y[p] = cast_bound(flow_uv[2 * p + 1], lower_bound, upper_bound);
        }
}

/* src/common.cpp:18-46. */
static uint8_t sat_u8(double v) {
    const int r = orc_cvround(v);
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

static double png_bound(double extent, double mn, double mx) {
    const double a = fabs(mn), b = fabs(mx);
    const double m = a > b ? a : b;               /* max(abs(min_v), abs(max_v)) */
    const double c = extent < m ? extent : m;     /* min(w, ...) */
    double bound = ceil((c * 128. / 127.) / 4) * 4;
    if (bound > 255. * 4)
        bound = 255. * 4;
    if ((int)bound % 8 == 0)
        bound += 4;
    return bound;
}

void orc_flow_to_png_planes(const float *flow_uv, int w, int h, uint8_t *img_x, uint8_t *img_y, double *bounds,
                            uint8_t *img_bgr) {
    const double base = 1. / 128.;
    const size_t n = (size_t)w * h;
    /* minMaxLoc: the search starts from +-infinity sentinels, so NaNs never win wherever they stand, and a plane without
     * a single comparable value reports min = max = 0 (upstream minMaxIdx: "if nothing was located, the values are 0") */
    double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (size_t p = 0; p < n; ++p) {
        const double u = flow_uv[2 * p], v = flow_uv[2 * p + 1];
        if (u < mnx) mnx = u;
        if (u > mxx) mxx = u;
        if (v < mny) mny = v;
        if (v > mxy) mxy = v;
    }
    if (mnx > mxx) mnx = mxx = 0.0; /* (an all-+inf / all--inf plane keeps its infinity: min <= max there) */
    if (mny > mxy) mny = mxy = 0.0;
    const double bound_x = png_bound((double)w, mnx, mxx), bound_y = png_bound((double)h, mny, mxy);
    const float ax = (float)(1. / (base * bound_x)), ay = (float)(1. / (base * bound_y));
    const int half = (int)((double)h / 2); /* Point(w - 1, half_h): double -> int by truncation; the box is inclusive */
    const uint8_t bx = sat_u8(bound_x / 4), by = sat_u8(bound_y / 4);
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const size_t p = (size_t)i * w + j;
            const float px = flow_uv[2 * p] * ax, py = flow_uv[2 * p + 1] * ay; /* convertTo: float product, float sum */
            const float vx = px + 128.f, vy = py + 128.f;
            const uint8_t qx = sat_u8((double)vx), qy = sat_u8((double)vy);
            if (img_x)
                img_x[p] = qx;
            if (img_y)
                img_y[p] = qy;
            if (img_bgr) {
                img_bgr[3 * p] = qx;
                img_bgr[3 * p + 1] = qy;
                img_bgr[3 * p + 2] = i <= half ? bx : by;
            }
        }
    if (bounds) {
        bounds[0] = bound_x;
        bounds[1] = bound_y;
    }
}
