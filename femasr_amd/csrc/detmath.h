// detmath.h — explicit elementary functions used by the HIP kernels.
//
// The parity contract (DESIGN.md "Arithmetic specification") needs exp / erf that
// give the same bits on gfx950 as in the CPU oracle, so they are written out with
// IEEE fp32 add / mul / fmaf only (no v_exp_f32 / v_rcp_f32 approximations):
//   det_expf : Cody-Waite range reduction, degree-7 Taylor, Horner with fmaf
//   det_erff : branch-free  sign(x) * (1 - exp(-a*Q(a))),  a = min(|x|,4),  Q = deg-9 fit of -ln(erfc(a))/a
// Build with -ffp-contract=off and correctly-rounded fp32 divide/sqrt (hipcc default).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float det_expf(float x)
{
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.98412698412698413e-4f;
    p = __builtin_fmaf(p, r, 1.38888888888888894e-3f);
    p = __builtin_fmaf(p, r, 8.33333333333333322e-3f);
    p = __builtin_fmaf(p, r, 4.16666666666666644e-2f);
    p = __builtin_fmaf(p, r, 1.66666666666666657e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const int ni = (int)n;
    return p * __uint_as_float((unsigned)(ni + 127) << 23);
}

__device__ __forceinline__ float det_erff(float x)
{
    const float ax = fminf(fabsf(x), 4.0f);
    float q = 0x1.19aba0p-21f;
    q = __builtin_fmaf(q, ax, -0x1.7c9856p-17f);
    q = __builtin_fmaf(q, ax, 0x1.b9a7a6p-14f);
    q = __builtin_fmaf(q, ax, -0x1.1711a6p-11f);
    q = __builtin_fmaf(q, ax, 0x1.6c55eep-10f);
    q = __builtin_fmaf(q, ax, 0x1.d916f4p-13f);
    q = __builtin_fmaf(q, ax, -0x1.3e7d62p-6f);
    q = __builtin_fmaf(q, ax, 0x1.a569bap-4f);
    q = __builtin_fmaf(q, ax, 0x1.45f0d6p-1f);
    q = __builtin_fmaf(q, ax, 0x1.20dd80p+0f);
    const float r = 1.0f - det_expf(-(ax * q));
    return copysignf(r, x);
}

// SiLU (fema_utils.py:54-55): x * sigmoid(x) = x / (1 + exp(-x)), IEEE division.
__device__ __forceinline__ float det_silu(float x) { return x / (1.0f + det_expf(-x)); }

// exact-erf GELU (network_swinir.py:15)
__device__ __forceinline__ float det_gelu(float x)
{
    return (0.5f * x) * (1.0f + det_erff(x * 0.707106781186547524f));
}

// ---- two-at-a-time versions on the packed fp32 ALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are IEEE operations per
// component, so each component is bit-identical to the scalar function above; the non-arithmetic steps - clamp, round,
// exponent insert, sign copy - stay scalar).  Used where the VALU shares its lanes with fp32 MFMA (GELU / LN in the
// linears): half the polynomial instructions.
typedef float det_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ det_f32x2 det_fma2(det_f32x2 a, det_f32x2 b, float c) { return __builtin_elementwise_fma(a, b, det_f32x2{c, c}); }

__device__ __forceinline__ det_f32x2 det_expf2(det_f32x2 x)
{
    x = det_f32x2{fminf(fmaxf(x[0], -87.0f), 88.0f), fminf(fmaxf(x[1], -87.0f), 88.0f)};
    const det_f32x2 t = x * det_f32x2{1.44269504088896341f, 1.44269504088896341f};
    const det_f32x2 n = {rintf(t[0]), rintf(t[1])};
    det_f32x2 r = __builtin_elementwise_fma(n, det_f32x2{-0.693359375f, -0.693359375f}, x);
    r = __builtin_elementwise_fma(n, det_f32x2{2.12194440e-4f, 2.12194440e-4f}, r);
    det_f32x2 p = {1.98412698412698413e-4f, 1.98412698412698413e-4f};
    p = det_fma2(p, r, 1.38888888888888894e-3f);
    p = det_fma2(p, r, 8.33333333333333322e-3f);
    p = det_fma2(p, r, 4.16666666666666644e-2f);
    p = det_fma2(p, r, 1.66666666666666657e-1f);
    p = det_fma2(p, r, 0.5f);
    p = det_fma2(p, r, 1.0f);
    p = det_fma2(p, r, 1.0f);
    const det_f32x2 sc = {__uint_as_float((unsigned)((int)n[0] + 127) << 23), __uint_as_float((unsigned)((int)n[1] + 127) << 23)};
    return p * sc;
}

// SiLU two at a time: the exponential polynomial on the packed ALU, the two IEEE divisions scalar (bit-identical per element)
__device__ __forceinline__ det_f32x2 det_silu2(det_f32x2 x)
{
    const det_f32x2 e = det_expf2(-x);
    const det_f32x2 d = det_f32x2{1.0f, 1.0f} + e;
    return det_f32x2{x[0] / d[0], x[1] / d[1]};
}

__device__ __forceinline__ det_f32x2 det_erff2(det_f32x2 x)
{
    const det_f32x2 ax = {fminf(fabsf(x[0]), 4.0f), fminf(fabsf(x[1]), 4.0f)};
    det_f32x2 q = {0x1.19aba0p-21f, 0x1.19aba0p-21f};
    q = det_fma2(q, ax, -0x1.7c9856p-17f);
    q = det_fma2(q, ax, 0x1.b9a7a6p-14f);
    q = det_fma2(q, ax, -0x1.1711a6p-11f);
    q = det_fma2(q, ax, 0x1.6c55eep-10f);
    q = det_fma2(q, ax, 0x1.d916f4p-13f);
    q = det_fma2(q, ax, -0x1.3e7d62p-6f);
    q = det_fma2(q, ax, 0x1.a569bap-4f);
    q = det_fma2(q, ax, 0x1.45f0d6p-1f);
    q = det_fma2(q, ax, 0x1.20dd80p+0f);
    const det_f32x2 e = det_expf2(-(ax * q));
    const det_f32x2 r = det_f32x2{1.0f, 1.0f} - e;
    return det_f32x2{copysignf(r[0], x[0]), copysignf(r[1], x[1])};
}

__device__ __forceinline__ det_f32x2 det_gelu2(det_f32x2 x)
{
    const det_f32x2 h = det_f32x2{0.5f, 0.5f} * x;
    const det_f32x2 e = det_erff2(x * det_f32x2{0.707106781186547524f, 0.707106781186547524f});
    return h * (det_f32x2{1.0f, 1.0f} + e);
}

