// farneback_kernels.h — device data layout and host-callable launchers of the -a=farn kernels
// (defined in farneback_kernels.hip).
//
// Layout in HBM (float32 planes, row pitch padded to 64 floats):
//   frame slot f : for every pyramid level k the polynomial expansion R of that frame, 5 planes
//                  (B.5) of pitch_k x h_k, at element offset r_off[k]; computed once per frame and
//                  used as R1 of pair i and R0 of pair i+step
//   pair slot b  : FARN_PL_COUNT planes sized for level 0: flow x/y (two sets: the iterations ping-pong
//                  between them, and a level is up-sampled from the set the previous level ended in) and,
//                  for the M-in-HBM kernels only (impl = 1, DFX_VAR_FARN_M_IN_HBM), M (two sets of 5 planes)
#pragma once

#include <hip/hip_runtime.h>

#include "dfx_device.h"

enum : int { FARN_PL_FX0 = 0, FARN_PL_FY0, FARN_PL_FX1, FARN_PL_FY1, FARN_PL_M0, FARN_PL_M1 = FARN_PL_M0 + 5,
             FARN_PL_COUNT = FARN_PL_M1 + 5 };

struct FarnPolyConsts { // B.3, non-negative halves
    float g[8], xg[8], xxg[8];
    float ig11, ig03, ig33, ig55;
};

struct FarnLevelGeom {
    int w, h, pitch;
    long long r_off; // element offset of this level's R (5 planes) inside a frame slot
};

struct FarnPairCtx {
    FarnLevelGeom L;
    const float *frame_R;   // base of frame slot 0
    long long frame_stride; // elements between frame slots
    float *planes;          // base of pair slot 0
    long long plane_stride, slot_stride;
    const PairDesc *pairs;
    int n_pairs;
};

void farn_launch_u8_to_f32(hipStream_t s, const unsigned char *src, long long src_frame_stride, long long src_pitch,
                           int n_frames, float *dst, long long dst_frame_stride, int w, int h, int pitch);
// vertical Gaussian pass at the 2 source rows every destination row samples (B.4 + E.1 fused)
void farn_launch_blur_v(hipStream_t s, const float *frames, long long frame_stride, int n_frames, int W, int H,
                        int pitch0, int dst_h, float ify, const float *ker_half, int half, float *tmpv,
                        long long tmpv_frame_stride, int skip_zero_weights);
// the same straight from the caller's 8-bit frames (the (float) of every load is E.2's convertTo: no u8 -> f32 pass)
void farn_launch_blur_v_u8(hipStream_t s, const unsigned char *frames, long long frame_stride, long long src_pitch,
                           int n_frames, int W, int H, int pitch0, int dst_h, float ify, const float *ker_half, int half,
                           float *tmpv, long long tmpv_frame_stride, int skip_zero_weights);
// horizontal Gaussian pass at the 2 source columns every destination pixel samples + bilinear resize
void farn_launch_blur_h_resize(hipStream_t s, const float *tmpv, long long tmpv_frame_stride, int n_frames, int W,
                               int H, int pitch0, int dst_w, int dst_h, int dst_pitch, float ifx, float ify,
                               const float *ker_half, int half, float *pyr, long long pyr_frame_stride,
                               int skip_zero_weights);
// polynomial expansion (B.5) of n_frames level images into their frame slots
void farn_launch_polyexp(hipStream_t s, const float *pyr, long long pyr_frame_stride, int n_frames,
                         const int *frame_slots, float *frame_R, long long frame_stride, FarnLevelGeom L,
                         FarnPolyConsts pc, int rows_per_workgroup);
// the defaults of the two frame-preparation switches (the engine overrides them from dfx_params.variant:
// DFX_VAR_FARN_EVAL_ZERO_TAPS, DFX_VAR_FARN_POLY_ONE_ROW)
int farn_skip_zero_weights_default();
int farn_polyexp_rows_default();
// flow(level k) = resize(flow(level k+1)) * (1/pyrScale), or zero at the coarsest level
void farn_launch_init_flow(hipStream_t s, const FarnPairCtx &c, int cur_set, int prev_w, int prev_h, int prev_pitch,
                           float ifx, float ify, float up, int zero);
void farn_launch_update_matrices(hipStream_t s, const FarnPairCtx &c, int flow_set, int m_set);
// boxFilter5 + updateFlow (+ updateMatrices) in one launch (B.8, B.9, B.7)
void farn_launch_iteration(hipStream_t s, const FarnPairCtx &c, int flow_set, int m_src, int half, float box_inv,
                           int do_matrices, int impl);
// the iteration with M recomputed where the box filter needs it (winSize 13 only): reads flow set flow_in, writes flow
// set flow_out; a stream down 64-column strips with a ring of 18 M rows in LDS (round 4's default)
// merged != nullptr (the last iteration of level 0): the new flow goes to the caller's interleaved (u, v) rows instead
void farn_launch_iter_stream(hipStream_t s, const FarnPairCtx &c, int flow_in, int flow_out, float box_inv, float *merged,
                             long long merged_stride);
// the first iteration of a level: its input flow is the coarser level's final flow (plane set prev_set, that level's
// geometry) up-sampled on the fly, or zero at the coarsest level — no init launch
void farn_launch_iter_stream_init(hipStream_t s, const FarnPairCtx &c, int prev_set, int flow_out, float box_inv,
                                  float *merged, long long merged_stride, int prev_w, int prev_h, int prev_pitch, float ifx,
                                  float ify, float up, int zero);
void farn_launch_merge(hipStream_t s, const FarnPairCtx &c, int flow_set, float *out, long long out_stride);
