// conv_common.h — definitions shared by the conv kernel translation units (kernels_conv.hip, kernels_conv_bf16.hip)
#pragma once
#include "common.h"

namespace {

struct ConvParams {
    const float *in, *w, *bias, *pro_a, *pro_b, *pro_c, *res1, *res2;
    float *out;
    const float *vq_zz, *vq_ee;
    float *vq_part;
    int vq_nblk;
    int B, H, W, Cin, Cout, ksz, stride, pad, up2, act, Ho, Wo;
    int M, K, nchunks, taps, MB, NB, NT32;
    int tilesX, tilesY;
};

constexpr int BK = 32;
constexpr int ALD = BK + 1;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

__device__ __forceinline__ float f4get(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
}


}  // namespace
