/*
 * femasr_hip_debug.h - test / measurement hooks of libfemasr_hip.so.  NOT part of the drop-in interface (include/femasr_hip.h):
 * nothing a host integration needs is declared here, and an installation may leave this header out.  tests/ and tools/ bind these
 * through femasr_amd/_lib.py.
 */
#ifndef FEMASR_HIP_DEBUG_H
#define FEMASR_HIP_DEBUG_H
#include "femasr_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Per-HANDLE test hook of the Winograd-form convs' size limits (kernels_wino.hip / kernels_wino_up2.hip address their tensors with
 * 32-bit byte offsets: a layer whose input or output has 2^31 or more elements, or 2^27 or more per image, runs in the direct /
 * phase-filter form instead).  log2_total / log2_image LOWER the two exponents for this handle's planner (31 / 27; smaller values
 * move the boundary down to sizes a test can allocate), 0 = the default.  Cached plans of the handle are dropped. */
int femasr_debug_set_wino_limits(femasr_handle *h, int log2_total, int log2_image);

/* The instruction behind linear_math 1: n independent v_mfma_f32_32x32x16_bf16 evaluations, case i = 16 products a[i][s] * b[i][s]
 * (bf16 bit patterns; k slot s = 8 * (lane / 32) + element) plus the fp32 accumulator input c[i] -> d[i].  tests/test_gpu_r5.py runs
 * constructed cases through it and compares them bit for bit with the oracle's restatement (orc_mfma_dot8). */
int femasr_debug_mfma_bf16(void *stream, const uint16_t *a, const uint16_t *b, const float *c, int n, float *d);

/* Tuning / test hook of the 1x1-conv and linear GEMM (kernels_gemm.hip): force one block configuration for every later launch -
 * 0: 128x128 tiles, 32-deep chunks, 2 stages;  1: 128x128, 16-deep, 3 stages;  2: 64x64 tiles (what small launches get);
 * any negative value: automatic choice by tile count (the default).  Results are bit-identical in every configuration (each
 * output is the same fmaf chain); the parity tests run all of them.  Returns the previous setting (negative = automatic). */
int femasr_gemm_force_config(int cfg);
/* Same kind of hook for the 3x3 / strided convs with more than 64 output channels: a launch of fewer than `blocks` 128-column
 * blocks runs with 64-column blocks instead (twice the blocks, half the serial chain each: batch-1 latency).  0 = never,
 * negative = the default (1.5 x the device's compute units: 384 on a 256-CU MI355X).  Bit-identical either way (same weights layout, same per-wave pixel tiles, same GroupNorm
 * partial-moment order).  Returns the previous threshold. */
int femasr_conv_small_launch_blocks(int blocks);

#ifdef __cplusplus
}
#endif
#endif /* FEMASR_HIP_DEBUG_H */
