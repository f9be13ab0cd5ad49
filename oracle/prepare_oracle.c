/*
 * oracle/prepare_oracle.c — TEST INFRASTRUCTURE ONLY.  See prepare_oracle.h (parity unpinned).
 * Written table-first like upstream's resize (offset / weight tables, then row passes), not per pixel.
 */
#include "prepare_oracle.h"

#include "oracle_common.h"

void orc_bgr2gray(const uint8_t *bgr, int w, int h, uint8_t *gray) {
    enum { RY15 = 9798, GY15 = 19235, BY15 = 3735, SHIFT = 15 };
    for (size_t i = 0; i < (size_t)w * h; ++i)
        gray[i] = (uint8_t)((bgr[3 * i] * BY15 + bgr[3 * i + 1] * GY15 + bgr[3 * i + 2] * RY15 + (1 << (SHIFT - 1))) >> SHIFT);
}

static short coef_to_short(float c) { /* saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE) */
    const int v = orc_cvround((double)(c * 2048.f));
    return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
}

void orc_resize_u8(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (sw == dw && sh == dh) { /* cv::resize copies */
        memcpy(dst, src, (size_t)sw * sh);
        return;
    }
    if (scale_x == 2.0 && scale_y == 2.0) { /* INTER_LINEAR -> INTER_AREA fast path */
        for (int dy = 0; dy < dh; ++dy) {
            const uint8_t *r0 = src + (size_t)(2 * dy) * sw, *r1 = r0 + sw;
            for (int dx = 0; dx < dw; ++dx)
                dst[(size_t)dy * dw + dx] = (uint8_t)((r0[2 * dx] + r0[2 * dx + 1] + r1[2 * dx] + r1[2 * dx + 1] + 2) >> 2);
        }
        return;
    }
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw), *ibeta = (short *)malloc(sizeof(short) * 2 * dh);
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx + 1 >= sw) {
            xmax = orc_imin(xmax, dx);
            if (sx >= sw - 1) {
                fx = 0;
                sx = sw - 1;
            }
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = coef_to_short(1.f - fx);
        ialpha[2 * dx + 1] = coef_to_short(fx);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = coef_to_short(1.f - fy);
        ibeta[2 * dy + 1] = coef_to_short(fy);
    }
    for (int dy = 0; dy < dh; ++dy) {
        const int sy0 = orc_imin(orc_imax(yofs[dy], 0), sh - 1), sy1 = orc_imin(orc_imax(yofs[dy] + 1, 0), sh - 1);
        const uint8_t *S0 = src + (size_t)sy0 * sw, *S1 = src + (size_t)sy1 * sw;
        for (int dx = 0; dx < xmax; ++dx) { /* HResizeLinear */
            const int sx = xofs[dx];
            row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx + 1] * ialpha[2 * dx + 1];
            row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx + 1] * ialpha[2 * dx + 1];
        }
        for (int dx = xmax; dx < dw; ++dx) {
            row0[dx] = S0[xofs[dx]] * 2048;
            row1[dx] = S1[xofs[dx]] * 2048;
        }
        const short b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        for (int dx = 0; dx < dw; ++dx) /* VResizeLinear<uchar, int, short> */
            dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs), free(yofs), free(ialpha), free(ibeta), free(row0), free(row1);
}

void orc_prepare_frame(const uint8_t *src, int sw, int sh, int channels, uint8_t *dst, int dw, int dh) {
    const uint8_t *gray = src;
    uint8_t *tmp = NULL;
    if (channels == 3) {
        tmp = (uint8_t *)malloc((size_t)sw * sh);
        orc_bgr2gray(src, sw, sh, tmp);
        gray = tmp;
    }
    orc_resize_u8(gray, sw, sh, dst, dw, dh);
    free(tmp);
}
