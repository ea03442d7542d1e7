// conv_common.h — definitions shared by the conv kernel translation units (kernels_conv.hip, kernels_conv_bf16.hip)
#pragma once
#include "common.h"

namespace {

struct ConvParams {
    const float *in, *w, *bias, *pro_a, *pro_b, *pro_c, *res1, *res2;
    const float *w_wino;  // Winograd-domain weights (femasr_repack_oihw_wino), kernels_wino.hip only
    const float *w_up2;   // 4 phase matrices of a nearest-x2 3x3 conv (femasr_repack_oihw_up2), halo kernels only
    float *out;
    const float *vq_zz, *vq_ee;
    float *vq_part;
    int vq_nblk;
    double *gn_part;      // fused GroupNorm partial moments of the OUTPUT, [B][tiles][32][2] (halo kernels) or null
    int B, H, W, Cin, Cout, ksz, stride, pad, up2, act, Ho, Wo;
    int M, K, nchunks, taps, MB, NB, NT32;
    int tilesX, tilesY;
    int kperm;            // conv_igemm: weights are in the GEMM layout of the 1x1 layers (k order 0,4,1,5,.. inside groups of 8)
};

constexpr int BK = 32;
constexpr int ALD = BK + 1;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Uniform base (SGPR pair) + 32-bit unsigned per-lane BYTE offset: the global_load/store "saddr + voffset" form.  The
// readfirstlane on both halves pins the (truly uniform) pointer into SGPRs and stops LLVM re-associating
// (base + lane offset) + uniform offset into per-element 64-bit VALU adds (which then stay live across batched loads).
typedef __attribute__((address_space(1))) const float *gcptr_f32;
typedef __attribute__((address_space(1))) float *gptr_f32;

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long a)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ float ldg_u32(const float *ubase, unsigned byteoff)
{
    const unsigned long long a = uniform_u64(reinterpret_cast<unsigned long long>(ubase));
    return *reinterpret_cast<gcptr_f32>(a + byteoff);
}
__device__ __forceinline__ void stg_u32(float *ubase, unsigned byteoff, float v)
{
    const unsigned long long a = uniform_u64(reinterpret_cast<unsigned long long>(ubase));
    *reinterpret_cast<gptr_f32>(a + byteoff) = v;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t ldg4_u32(const float *ubase, unsigned byteoff)
{
    const unsigned long long a = uniform_u64(reinterpret_cast<unsigned long long>(ubase));
    return *reinterpret_cast<__attribute__((address_space(1))) const f32x4_t *>(a + byteoff);
}
__device__ __forceinline__ void stg4_u32(float *ubase, unsigned byteoff, f32x4_t v)
{
    const unsigned long long a = uniform_u64(reinterpret_cast<unsigned long long>(ubase));
    *reinterpret_cast<__attribute__((address_space(1))) f32x4_t *>(a + byteoff) = v;
}
constexpr int TPITCH = 36;                   // epilogue transpose scratch: 32 rows x 36 floats per wave (16-byte rows)
constexpr int TSCRATCH = 32 * TPITCH;        // floats per wave

__device__ __forceinline__ float f4get(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// 32-column weight tile a wave reads for its accumulator tile t: tiles past the packed matrix (Cout not a multiple of the
// block width) are clamped to the last real tile - their products land in columns >= Cout, which are never stored - so
// the fragment loads never leave the packed buffer.
__device__ __forceinline__ int wtile(int n0, int t, int nt32)
{
    const int idx = (n0 >> 5) + t;
    return idx < nt32 ? idx : nt32 - 1;
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
}


}  // namespace
