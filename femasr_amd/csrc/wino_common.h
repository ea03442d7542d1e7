// wino_common.h - launch parameters, LDS geometry, the 6-point transforms and the experiment switches of the F(4x4,3x3) kernel
// (kernels_wino.hip: 2 x 16x16 pixels x 64 output channels per block).  Round 4's second block shape (16x16 pixels x 128 channels,
// kernels_wino_c128.hip: 7 % fewer cycles, 3 % more time at the 1400 W package limit) was removed in round 6; git history and profiles/r04_j_* have it.
// The including file defines FEMASR_WTT_BUF (the name of its cycle-stamp buffer) first.
#pragma once
#include "conv_common.h"
#include "detmath.h"
#include <stdlib.h>
#include <type_traits>
#include <atomic>
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float tf2 __attribute__((ext_vector_type(2)));

#ifdef FEMASR_WINO_TT      // tools/build_debug.sh tt: cycle stamps (s_memtime) of waves 0 and 7 of every block, written straight to a global
// buffer [block][wave 0 | 7][128] - no accumulators in registers, one scalar read + one 8-byte store per stamp.  Slots: 63 block start,
// 0 prologue done, 16 + s main-loop step s done (s < 40), 1 main loop done, per epilogue round r: 2 + 4r accumulators in LDS,
// 3 + 4r residuals requested + barrier passed, 4 + 4r round computed / stored, 5 + 4r second barrier; 10 end; inside step s < 24:
// 64 + 2s M phase issued (+ hoisted transform reads), 65 + 2s T phase done (before the barrier).
__device__ unsigned long long *FEMASR_WTT_BUF;
#define WTT(slot) { if (lane == 0 && (wave == 0 || wave == 7)) FEMASR_WTT_BUF[((size_t)blockIdx.x * 2 + (wave == 7 ? 1 : 0)) * 128 + (slot)] = __builtin_readcyclecounter(); }
#define WTTR(slot) { if (lane == 0 && (wave == 0 || wave == 7)) FEMASR_WTT_BUF[((size_t)blockIdx.x * 2 + (wave == 7 ? 1 : 0)) * 128 + (slot)] = wall_clock64(); }
#define WTT_INIT { WTTR(62) WTT(63) }
#define WTT_END { WTT(10) WTTR(11) }
#else
#define WTT(slot) {}
#define WTT_INIT
#define WTT_END {}
#endif

namespace {

struct WinoParams {
    const float *in, *u, *bias, *pro_a, *pro_b, *res1, *res2;
    float *out;
    double *gn_part;
    int B, H, W, Cin, Cout;
    int sbX, sbY, nsb;       // 16x16-pixel sub-blocks per row / per column / in the batch
    int MB, NB, nsteps, NT32;
};

constexpr int W4_NT = 512;
constexpr int W4_PS = 10;                         // floats per patch pixel: 8 channels + 2 (tiles 4 pixels apart land 8 banks apart)
constexpr int W4_PW = 18;
constexpr int W4_PPIX = W4_PW * W4_PW;            // 324
constexpr int W4_PSZ = 2 * W4_PPIX * W4_PS;       // floats per patch buffer (two sub-blocks)
constexpr int W4_VSZ = 36 * 32 * 8;               // floats per V buffer
constexpr int W4_MAIN = 2 * W4_PSZ + 2 * W4_VSZ;  // main-loop floats ahead of the GN coefficient table
constexpr int W4_MX = 36 * 32 * 32;               // epilogue: one 32-column tile of all components
constexpr int W4_RED = 8 * 2 * 16 * 2 * 2;        // floats: [8 waves][2 sub-blocks][<= 16 groups][2] doubles
constexpr int W4_UNITS_SB = W4_PPIX * 2;          // float4 staging units per sub-block and step (648)

inline size_t wino_lds_bytes(int cin, bool gn)
{
    const size_t main_f = (size_t)W4_MAIN + (gn ? 4 * (size_t)cin : 0), epi_f = (size_t)W4_MX + W4_RED;
    return (main_f > epi_f ? main_f : epi_f) * sizeof(float);
}

// 6-point transforms of F(4x4,3x3).  Input rows of B^T (Lavin & Gray), written so that each value is one fixed sequence of
// IEEE operations (the oracle restates them literally):
//   r0 = 4 d0 - 5 d2 + d4          r1 = (d4 - 4 d2) + (d3 - 4 d1)      r2 = (d4 - 4 d2) - (d3 - 4 d1)
//   r3 = (d4 - d2) + 2 (d3 - d1)   r4 = (d4 - d2) - 2 (d3 - d1)        r5 = 4 d1 - 5 d3 + d5
__device__ __forceinline__ void bt_lo(float d0, float d1, float d2, float d3, float d4, float &r0, float &r1, float &r2)
{
    r0 = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
    const float a = __builtin_fmaf(-4.0f, d2, d4), b = __builtin_fmaf(-4.0f, d1, d3);
    r1 = a + b;
    r2 = a - b;
}
__device__ __forceinline__ void bt_hi(float d1, float d2, float d3, float d4, float d5, float &r3, float &r4, float &r5)
{
    const float c = d4 - d2, e = d3 - d1;
    r3 = __builtin_fmaf(2.0f, e, c);
    r4 = __builtin_fmaf(-2.0f, e, c);
    r5 = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
}
// output rows of A^T:  y0 = (m0 + (m1 + m2)) + (m3 + m4),  y1 = (m1 - m2) + 2 (m3 - m4),  y2 = (m1 + m2) + 4 (m3 + m4),
//                      y3 = ((m1 - m2) + 8 (m3 - m4)) + m5
// (on PAIRS: v_pk_add_f32 / v_pk_fma_f32 are IEEE per component)
__device__ __forceinline__ void at6(tf2 m0, tf2 m1, tf2 m2, tf2 m3, tf2 m4, tf2 m5, tf2 &y0, tf2 &y1, tf2 &y2, tf2 &y3)
{
    const tf2 pp = m1 + m2, qq = m1 - m2, rr = m3 + m4, ss = m3 - m4;
    y0 = (m0 + pp) + rr;
    y1 = __builtin_elementwise_fma(tf2{2.0f, 2.0f}, ss, qq);
    y2 = __builtin_elementwise_fma(tf2{4.0f, 4.0f}, rr, pp);
    y3 = __builtin_elementwise_fma(tf2{8.0f, 8.0f}, ss, qq) + m5;
}

// G g G^T per (o, i), rows (over ky) then columns (over kx), each one fixed sequence of IEEE operations:
//   u0 = g0 * 0.25     u1 = ((g0 + g1) + g2) * (-1/6)     u2 = ((g0 - g1) + g2) * (-1/6)
//   u3 = fma(4, g2, fma(2, g1, g0)) * (1/24)     u4 = fma(4, g2, fma(-2, g1, g0)) * (1/24)     u5 = g2
__device__ __forceinline__ float wino_g6(int r, float g0, float g1, float g2)
{
    const float c6 = -1.0f / 6.0f, c24 = 1.0f / 24.0f;
    switch (r) {
    case 0: return g0 * 0.25f;
    case 1: return ((g0 + g1) + g2) * c6;
    case 2: return ((g0 - g1) + g2) * c6;
    case 3: return __builtin_fmaf(4.0f, g2, __builtin_fmaf(2.0f, g1, g0)) * c24;
    case 4: return __builtin_fmaf(4.0f, g2, __builtin_fmaf(-2.0f, g1, g0)) * c24;
    default: return g2;
    }
}

}  // namespace

// Schedule variants built and A/B-measured in round 4 (all bit-identical; profiles/r04_wino_variants.txt), removed again except the last two:
//   * the patch of step s+2 requested at the start of the M phase instead of at pair 5: 4 % SLOWER - loads return in order, the slow HBM
//     request ahead of the L2-resident U fragments delays them;
//   * staging ahead of the transform arithmetic (its LDS reads in flight meanwhile): 0 .. +1 %;
//   * the transform's patch reads at the start of the M phase, or the whole transform inside the M phase: +-1 %;
//   * the input transform on (tile, channel PAIR) items - both passes packed, 72 instead of 2 x 57 instructions per SIMD and step, but
//     only on waves 0-3: 2 - 9 % SLOWER: one wave's VALU stream alone does not reach the issue rate two interleaved waves do;
//   * prologue: every global request first, the patches of steps 0 and 1 requested together (second register set): 1 % faster - kept;
//   * FEMASR_WINO_DEEP (below): the second register set also used in the main loop - see the main loop.
#ifndef FEMASR_WINO_DEEP
#define FEMASR_WINO_DEEP 1
#endif
#ifndef FEMASR_WINO_NT       // experiment: cache-policy bits of the streaming accesses (bit 1 = nt): 1 = input patches, 2 = residual loads / output stores
#define FEMASR_WINO_NT 0
#endif
#define W_NT_IN ((FEMASR_WINO_NT & 1) ? 2 : 0)
#define W_NT_IO ((FEMASR_WINO_NT & 2) ? 2 : 0)
#ifndef FEMASR_WINO_ABL      // ablation experiments (tools/build_debug.sh): bit 0 no patch loads, 1 no U loads, 2 no transform, 3 no MFMAs,
#define FEMASR_WINO_ABL 0    // 4 no output items, 5 no activation, 6 no staging stores
#endif
