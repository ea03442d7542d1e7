// common.h — internal declarations shared by the .hip translation units of libfemasr_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/femasr_hip_debug.h"      // (includes femasr_hip.h)

// thread-local error message plumbing (model.hip)
int femasr_set_error(int code, const char *fmt, ...);
#define FEMASR_CHECK_HIP(expr)                                                            \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return femasr_set_error(FEMASR_ERR_HIP, "%s failed: %s (%s:%d)", #expr,       \
                                    hipGetErrorString(_e), __FILE__, __LINE__);           \
    } while (0)
#define FEMASR_REQUIRE(cond, ...)                                                         \
    do {                                                                                  \
        if (!(cond)) return femasr_set_error(FEMASR_ERR_INVALID, __VA_ARGS__);            \
    } while (0)

// conv launcher with the VQ-argmin epilogue option (kernels_conv.hip)
//   vq_part != nullptr : instead of storing the tile, each block writes, per row, the
//   first-min (distance, column) over its BN columns of d = (vq_zz[row] + vq_ee[col]) - 2*acc
//   to vq_part[(row*vq_nblk + nblock)*2 + {0,1}] (column stored as a float-bit-cast int).
struct conv_vq_epilogue {
    const float *zz;
    const float *ee;
    float *part;
    int nblk;
};
int femasr_conv2d_launch(hipStream_t s, const femasr_conv_args *a, const conv_vq_epilogue *vq,
                         int *variant_out, double *flops_out);
int femasr_conv_variant_count();
const char *femasr_conv_variant_name(int v);
bool femasr_conv_halo_eligible(const femasr_conv_args *a);      // 3x3 s1 p1, Cin % 32 == 0: the halo kernels
// fused GroupNorm(32) partial moments in a halo conv's epilogue: channels per group a power of two <= 32
inline bool femasr_gn_fusable(int cout) { const int cg = cout / 32; return cout % 32 == 0 && cg >= 1 && cg <= 32 && (cg & (cg - 1)) == 0; }

// convs with <= 4 output channels and a 3x3 stride-1 halo shape (out_conv) keep a compact [k][4] weight copy behind the
// fragment-major matrix for the direct VALU kernel (an MFMA tile would compute 32 columns for 3)
inline size_t femasr_compact_weight_floats(int O, int I, int kh, int kw)
{
    return (O <= 4 && kh == 3 && kw == 3 && (I % 32) == 0) ? (size_t)I * 9 * 4 : 0;
}

// 1x1 convs / nn.Linear / VQ distance matrix on the LDS-DMA GEMM (kernels_gemm.hip)
bool femasr_gemm_eligible(const femasr_conv_args *a);
int femasr_gemm_launch(hipStream_t s, const femasr_conv_args *a, const conv_vq_epilogue *vq, int *variant_out, double *flops_out);
int femasr_gemm_variant_count();
const char *femasr_gemm_variant_name(int v);
int femasr_repack_k1(hipStream_t s, const float *in, int O, int I, float *out);

// 1x1 convs / nn.Linear as an fp32-grade product on the bf16 matrix pipe (kernels_gemm_bf16.hip)
bool femasr_gemm_bf16s_shape_ok(const femasr_conv_args *a);
bool femasr_conv3x3_bf16s_shape_ok(const femasr_conv_args *a);      // the 3x3 stride-1 pad-1 form (K = 9 Cin) of the same kernel
int femasr_gemm_bf16s_launch(hipStream_t s, const femasr_conv_args *a, const void *w_bf16s, int *variant_out, double *flops_out);
int femasr_gemm_bf16s_variant_count();
const char *femasr_gemm_bf16s_variant_name(int v);

// Winograd F(4x4,3x3) 3x3 convs (kernels_wino.hip)
constexpr int FEMASR_WINO_LOG2_TOTAL = 31, FEMASR_WINO_LOG2_IMAGE = 27;      // element limits of the Winograd-form kernels (32-bit byte offsets)
bool femasr_conv_wino_shape_ok(const femasr_conv_args *a);
bool femasr_conv_wino_shape_ok_lim(const femasr_conv_args *a, int log2_total, int log2_image);      // a handle's planner may lower the limits (femasr_debug_set_wino_limits)
int femasr_conv_wino_gn_tiles(int H, int W);      // fused GroupNorm partials of a Winograd conv: one per 16x16-pixel sub-block
int femasr_conv_wino_launch(hipStream_t s, const femasr_conv_args *a, int *variant_out, double *flops_out);
int femasr_conv_wino_variant_count();
const char *femasr_conv_wino_variant_name(int v);
// nn.Upsample(x2) + 3x3 conv in the 25-product Winograd-type form (kernels_wino_up2.hip); GroupNorm partials per 16x16 OUTPUT sub-block
bool femasr_conv_wino_up2_shape_ok(const femasr_conv_args *a);
bool femasr_conv_wino_up2_shape_ok_lim(const femasr_conv_args *a, int log2_total, int log2_image);
int femasr_conv_wino_up2_launch(hipStream_t s, const femasr_conv_args *a, double *flops_out);
const char *femasr_conv_wino_up2_variant_name();

// bf16x3 3x3 halo convs (kernels_conv_bf16.hip)
bool femasr_conv_bf16x3_eligible(const femasr_conv_args *a);
bool femasr_conv_bf16x3_shape_ok(const femasr_conv_args *a);      // the same rule without the w_bf16x3 pointer (planner)
int femasr_conv_bf16x3_launch(hipStream_t s, const femasr_conv_args *a, int *variant_out, double *flops_out);
int femasr_conv_bf16x3_variant_count();
const char *femasr_conv_bf16x3_variant_name(int v);
