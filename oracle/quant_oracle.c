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
