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
