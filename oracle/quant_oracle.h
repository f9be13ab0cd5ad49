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

#ifdef __cplusplus
}
#endif
#endif
