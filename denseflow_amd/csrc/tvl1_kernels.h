// tvl1_kernels.h — host-callable launchers of the TVL1 kernels (defined in tvl1_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "dfx_device.h"

void tvl1_launch_u8_to_f32(hipStream_t s, const unsigned char *src, long long src_frame_stride, long long src_pitch,
                           const int *frame_slots, int n_frames, float *dst, long long dst_frame_stride, int w, int h,
                           int pitch);
void tvl1_launch_pyr_down(hipStream_t s, float *frame_I, long long frame_stride, const int *frame_slots, int n_frames,
                          long long src_off, int sw, int sh, int spitch, long long dst_off, int dw, int dh, int dpitch,
                          float ifx, float ify);
void tvl1_launch_centered_gradient(hipStream_t s, const float *frame_I, float *frame_Ix, float *frame_Iy,
                                   long long frame_stride, const int *frame_slots, int n_frames, long long off, int w,
                                   int h, int pitch);
void tvl1_launch_level_begin(hipStream_t s, const Tvl1LevelCtx &c, int first_level);
void tvl1_launch_warp(hipStream_t s, const Tvl1LevelCtx &c, int step_id); // dedicated backward-warp kernel of a step
// the warp AND the head of the loop it starts (tvl1_head_kernels.hip), in place of tvl1_launch_warp
void tvl1_launch_warp_head(hipStream_t s, const Tvl1LevelCtx &c, int step_id, int math);
int tvl1_head_blocks(const Tvl1LevelCtx &c); // workgroups per pair of that launch
void tvl1_launch_step(hipStream_t s, const Tvl1LevelCtx &c, int step_id, int impl, int math);
int tvl1_step_blocks(const Tvl1LevelCtx &c, int impl); // workgroups per pair of a step launch
int tvl1_fused_max_k();                                // largest supported inner-iteration fusion
void tvl1_launch_upsample_u(hipStream_t s, const Tvl1LevelCtx &c_src, int dw, int dh, int dpitch, float ifx, float ify,
                            float up);
void tvl1_launch_merge(hipStream_t s, const Tvl1LevelCtx &c0, float *out, long long out_stride);
