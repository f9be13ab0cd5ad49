/*
 * oracle/quant_oracle.h — TEST INFRASTRUCTURE ONLY (see oracle_common.h for who may use oracle/).
 *
 * CPU restatement of the reference's flow bounding, convertFlowToImage,
 * /root/reference/src/common.cpp:4-16, as encodeFlowMap (:48-64) calls it with
 * (lowerBound, higherBound) = (-bound, +bound).
 *
 * PARITY PINNED for this function: unlike the optical-flow algorithms, its source is in the
 * reference repository itself.  oracle/Makefile target `ref` compiles those very lines (piped from
 * /root/reference, never copied into this repository) against a 30-line stand-in for cv::Mat /
 * cvRound (ref_shim.h) into oracle/_ref/libref_quant.so; tests/test_oracle_quant.py checks this
 * restatement against it and against tests/golden/quant_golden.npz, which was generated from it.
 */
#ifndef DFX_QUANT_ORACLE_H
#define DFX_QUANT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flow_uv: h rows of w interleaved (u, v) floats, dense.  img_x / img_y: h*w bytes each, dense. */
void orc_flow_to_u8(const float *flow_uv, int w, int h, double lower_bound, double upper_bound, uint8_t *img_x,
                    uint8_t *img_y);

/* The -st=png scheme, convertFlowToPngImage, /root/reference/src/common.cpp:18-46 (PARITY PINNED the same way:
 * oracle/_ref/libref_png.so is those lines compiled against ref_png_shim.h; tests/golden/png_planes_golden.npz was
 * minted from it).  Per flow: bound_x = min(1020, ceil((min(w, max|u|) * 128 / 127) / 4) * 4), + 4 when that integer is a
 * multiple of 8 (bound_y with h and v); plane x = saturate_u8(u * (float)(1 / (bound_x / 128)) + 128.f) (float product,
 * float sum, round half to even), plane y likewise; the third channel carries bound_x / 4 on rows 0 .. int(h / 2)
 * and bound_y / 4 below.  img_x / img_y: h*w bytes; bounds[2] = {bound_x, bound_y}.  img_bgr (may be NULL): h*w*3. */
void orc_flow_to_png_planes(const float *flow_uv, int w, int h, uint8_t *img_x, uint8_t *img_y, double *bounds,
                            uint8_t *img_bgr);

#ifdef __cplusplus
}
#endif
#endif
