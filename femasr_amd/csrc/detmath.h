// detmath.h — explicit elementary functions used by the HIP kernels.
//
// The parity contract (DESIGN.md "Arithmetic specification") needs exp / erf that
// give the same bits on gfx950 as in the CPU oracle, so they are written out with
// IEEE fp32 add / mul / fmaf only (no v_exp_f32 / v_rcp_f32 approximations):
//   det_expf : Cody-Waite range reduction, degree-7 Taylor, Horner with fmaf
//   det_erff : |x|<1  x*P(x^2) (deg 6);  1<=|x|<4  1-exp(Q(|x|-2.5)) (deg 8);  else 1
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
    const float ax = fabsf(x);
    float r;
    if (ax < 1.0f) {
        const float t = ax * ax;
        float p = 0x1.4fd5fap-14f;
        p = __builtin_fmaf(p, t, -0x1.a63fe2p-11f);
        p = __builtin_fmaf(p, t, 0x1.545368p-8f);
        p = __builtin_fmaf(p, t, -0x1.b80286p-6f);
        p = __builtin_fmaf(p, t, 0x1.ce2d7cp-4f);
        p = __builtin_fmaf(p, t, -0x1.812740p-2f);
        p = __builtin_fmaf(p, t, 0x1.20dd76p+0f);
        r = p * ax;
    } else if (ax < 4.0f) {
        const float u = ax - 2.5f;
        float q = 0x1.b49612p-20f;
        q = __builtin_fmaf(q, u, -0x1.bdf536p-17f);
        q = __builtin_fmaf(q, u, 0x1.4467eap-14f);
        q = __builtin_fmaf(q, u, -0x1.b89f56p-12f);
        q = __builtin_fmaf(q, u, 0x1.1bccd0p-9f);
        q = __builtin_fmaf(q, u, -0x1.63cf4ep-7f);
        q = __builtin_fmaf(q, u, -0x1.e34608p-1f);
        q = __builtin_fmaf(q, u, -0x1.569252p+2f);
        q = __builtin_fmaf(q, u, -0x1.f3a2dcp+2f);
        r = 1.0f - det_expf(q);
    } else {
        r = 1.0f;
    }
    return copysignf(r, x);
}

// SiLU (fema_utils.py:54-55): x * sigmoid(x) = x / (1 + exp(-x)), IEEE division.
__device__ __forceinline__ float det_silu(float x) { return x / (1.0f + det_expf(-x)); }

// exact-erf GELU (network_swinir.py:15)
__device__ __forceinline__ float det_gelu(float x)
{
    return (0.5f * x) * (1.0f + det_erff(x * 0.707106781186547524f));
}
