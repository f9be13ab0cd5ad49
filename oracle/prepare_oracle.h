/*
 * oracle/prepare_oracle.h — TEST INFRASTRUCTURE ONLY (see oracle_common.h for who may use oracle/).
 *
 * CPU restatement of the frame preparation the reference does per frame in
 * DenseFlow::load_frames_batch, /root/reference/src/denseflow_gpu.cpp:146-177:
 *     cvtColor(capture_frame, frame_gray, COLOR_BGR2GRAY);      (:163)
 *     cv::resize(frame_gray, resized_frame_gray, size);          (:169, default INTER_LINEAR)
 *
 * PARITY UNPINNED: both routines live in OpenCV imgproc 4.5.2 (pinned by
 * /root/reference/docker/Dockerfile:6), which is not vendored under /root/reference and cannot be
 * built here; the reference has no test for them.  This file restates the published 8-bit paths:
 *   cvtColor BGR2GRAY (color_rgb: RGB2Gray<uchar>): (B*BY15 + G*GY15 + R*RY15 + 2^14) >> 15 with
 *       RY15 = 9798, GY15 = 19235, BY15 = 3735;
 *   cv::resize (resize.cpp): scale = 1/(dsize/ssize) in double; an INTER_LINEAR request whose scale is
 *       exactly 2 in both directions is executed as INTER_AREA (2x2 average, (sum + 2) >> 2);
 *       otherwise per-axis tables — fx = (float)((d + 0.5)*scale - 0.5), s = floor(fx), fx -= s;
 *       along x: s < 0 -> (s, fx) = (0, 0); s >= width-1 -> (width-1, 0); along y the table keeps fx
 *       and the row loop clamps the two row indices; weights saturate_cast<short>(w * 2048);
 *       HResizeLinear: D = S[s]*a0 + S[s+1]*a1 (int); VResizeLinear<uchar,int,short>:
 *       dst = (((b0*(S0 >> 4)) >> 16) + ((b1*(S1 >> 4)) >> 16) + 2) >> 2.
 * It is cross-checked against an independent NumPy restatement (tests/numpy_restatement.py) and
 * known answers (identity, constants, 2x decimation, monotone ramps).
 */
#ifndef DFX_PREPARE_ORACLE_H
#define DFX_PREPARE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bgr: h rows of w interleaved (B, G, R) bytes, dense.  gray: h*w bytes, dense. */
void orc_bgr2gray(const uint8_t *bgr, int w, int h, uint8_t *gray);

/* cv::resize(src, dst, Size(dw, dh)) for CV_8UC1, default interpolation.  Dense buffers. */
void orc_resize_u8(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);

/* The loader's sequence for one frame: gray conversion when channels == 3, then the resize when the size
 * differs (the reference only calls cv::resize when a new size was requested, :167-172). */
void orc_prepare_frame(const uint8_t *src, int sw, int sh, int channels, uint8_t *dst, int dw, int dh);

#ifdef __cplusplus
}
#endif
#endif
