// kernels_conv_bf16.hip — 3x3 halo convolution on the bf16 matrix cores with a 3-term split ("bf16x3").
//
// Where it is used: ONLY for convs that cannot move a VQ index - behind the codebook lookup (after_quant conv, the three
// DecoderBlocks, out_conv; femasr_arch.py:195-211,267-273,352,366-369) and the encoder's up-blocks whose outputs are
// only the decoder's skip features - and only when the caller opts in (femasr_set_decoder_math(1)).
// Everything that feeds the argmin stays on the exact-fp32 kernels (kernels_conv.hip), because VQ index parity
// needs fp32-grade z (hard part 1); after the lookup the contract is the north-star's 1e-3 max-abs.
//
// Arithmetic: x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi) for activations AND weights;
//   x*w ~= hi_x*hi_w + hi_x*lo_w + lo_x*hi_w          (the dropped lo*lo term is 2^-16 relative)
// three v_mfma_f32_32x32x16_bf16 per 16-deep k-step, fp32 accumulation: relative error ~2e-5 per product term,
// i.e. ~1e-5 of the activation scale after accumulation — two orders below the 1e-3 bound (measured in
// tests/test_gpu_kernels.py / test_gpu_network.py).  16x the fp32 MFMA rate / 3 passes = 5.3x per K-step.
//
// Structure = conv3x3_halo (kernels_conv.hip): 8x16 output pixels x BN channels per block, one halo patch per
// 32-channel block staged ONCE (GroupNorm-apply + SiLU in fp32, then split) and swept by all 9 taps; weights are
// pre-split and stored fragment-major by femasr_repack_oihw_bf16x3 ([q][ntile][k-step][hi|lo][lane] x 8 bf16: every
// wave-level load is one contiguous KiB) and read straight into MFMA operands.
//   LDS patch image: ushort [buffers][hi|lo][pixel][40]  (80-byte pixel pitch -> the ds_read_b128 of the A fragment
//   A[i=lane&31][k=8*(lane>>5)..+7] is conflict-free across each 16-lane group), + the sample's GN coefficients.
// Blocks: 256 threads = 4 waves.  Cout > 128: one 256-wide block, 128 px x 64 ch per wave (rotating A fragments);
//   65..128: 64 x 64 per wave; 33..64: 64 px x 32 ch per wave, SINGLE-buffered patch (4 blocks per CU); <= 32: 32 x 32.
// The main loop is unconditional straight-line code (9 taps unrolled) so every s_waitcnt is exact; the epilogue
// transposes tiles through LDS for dwordx4 stores and can emit per-tile GroupNorm partial moments of its output.
#include "conv_common.h"
#include "detmath.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef FEMASR_TAPTIME
// debug build only (tools/build_debug.sh): per-wave cycle sums (tap 0..8, barrier, total, prologue, epilogue).  Level 1
// stamps only prologue / main loop / epilogue (4 s_memtime per wave: negligible intrusion); FEMASR_TAPTIME=2 also
// stamps every tap and barrier (each stamp drains lgkmcnt, so that mode slows the kernel and is only for ratios).
__device__ unsigned long long g_taptime[16 * 65536];     // [wave slot][16], plain stores (same-address atomics serialise)
#define TT_STAMP_ALWAYS(slot)                                       \
    {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tt[slot] += now_ - tprev;                                    \
        tprev = now_;                                                \
    }
#if FEMASR_TAPTIME >= 2
#define TT_STAMP(slot) TT_STAMP_ALWAYS(slot)
#else
#define TT_STAMP(slot) {}
#endif
__constant__ int g_dbg16_flags;     // debug build only: 1 = skip the prologue's residual loads, 2 = skip its first patch loads
#define DBG16_ON(bit) (g_dbg16_flags & (bit))
#else
#define TT_STAMP(slot) {}
#define TT_STAMP_ALWAYS(slot) {}
#define DBG16_ON(bit) false
#endif

namespace {

constexpr int PPITCH = 40;     // ushorts per patch pixel (32 channels + 8 pad)

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &lo)
{
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
}

// SiLU with the hardware exp2 / rcp approximations (1 ulp each): this path is tolerance-based (1e-3 contract, ~1e-5
// measured), so the 28-instruction bit-reproducible det_silu of the fp32 kernels is not needed here.
__device__ __forceinline__ float fast_silu(float t)
{
    return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.44269504088896341f));
}

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 &v) { return __builtin_bit_cast(bf16x8, v); }

template <int BN, int WM, int WN, int PRO, bool UP2>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8 || (BN <= 64 && !UP2)) ? 4 : 2) void conv3x3_halo_bf16x3_kernel(const ConvParams p, const uint4 *__restrict__ wsplit,
                                                                                          double *__restrict__ stats_part)
{
    constexpr int BM = 128, TW = 16, NT = WM * WN * 64;
    constexpr int PH = UP2 ? 6 : 10, PW = UP2 ? 10 : 18, PP = PH * PW;
    constexpr int PUNITS = (PP * 8 + NT - 1) / NT, PROWS = NT / 8;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1, "tile config");
    static_assert(PRO != FEMASR_PRO_LN, "no LayerNorm prologue on 3x3 convs");
    static_assert(1 + PUNITS <= 9, "patch slices are spread over taps 1..");
    // ROT: the 128 px x 64 ch wave tile (8 accumulator tiles = 128 registers) leaves no room for the two-deep A-fragment
    // and whole-patch staging registers of the smaller tiles: A fragments rotate through ONE hi and ONE lo set (lo(s)
    // is fetched under the two hi terms, hi(s+1) under the lo term) and the next patch is staged in two halves.
    constexpr bool ROT = TM * TN >= 8;
    // SB: single-buffered patch for the narrow layers (Cout <= 64: the 288^2 / 576^2 decoder end).  They are HBM-side with
    // only 2-4 channel blocks per tile; halving the LDS footprint doubles the resident blocks (4 per CU) and with them the
    // bytes in flight, which matters more there than overlapping the patch stores with this block's own MFMAs.
    constexpr bool SB = BN <= 64 && !UP2;
    static_assert(!(SB && ROT), "single-buffered patch is for the narrow tilings");
    constexpr int NG = ROT ? 3 : 1, GS = (PUNITS + NG - 1) / NG;      // patch units are loaded / stored in NG groups of GS

    extern __shared__ __attribute__((aligned(16))) unsigned short smem_u16[];
    constexpr int HALF = (PP + 1) * PPITCH;      // ushorts per (buffer, hi|lo) image; pixel PP is a write-only dummy slot
    unsigned short *Ps = smem_u16;               // [2][2][PP][PPITCH]

    const int t = threadIdx.x, lane = t & 63;
#ifdef FEMASR_TAPTIME
    unsigned long long tt[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#endif
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB;
    int tile = L / p.NB;
    const int tx = tile % p.tilesX;
    tile /= p.tilesX;
    const int ty = tile % p.tilesY;
    const int n = tile / p.tilesY;
    const int oy0 = ty * 8, ox0 = tx * TW, n0 = nb * BN;
    const int sy0 = UP2 ? (oy0 >> 1) - 1 : oy0 - 1, sx0 = UP2 ? (ox0 >> 1) - 1 : ox0 - 1;

    const int kq = t & 7;
    unsigned poff[PUNITS];
    unsigned pmask = 0;
#pragma unroll
    for (int i = 0; i < PUNITS; ++i) {
        const int pix = (t >> 3) + PROWS * i;
        const int ppy = pix / PW, ppx = pix - ppy * PW;
        const int sy = sy0 + ppy, sx = sx0 + ppx;
        const bool ok = (pix < PP) & (sy >= 0) & (sy < p.H) & (sx >= 0) & (sx < p.W);
        poff[i] = ok ? (unsigned)((((size_t)n * p.H + sy) * p.W + sx) * p.Cin + 4 * kq) : 0u;
        pmask |= (ok ? 1u : 0u) << i;
    }

    // GroupNorm coefficients of this sample: staged once in LDS behind the patch buffers ([2][Cin] floats), read back per
    // unit at store time (no per-channel-block global loads in the main loop, no registers held across taps)
    float *gco = reinterpret_cast<float *>(smem_u16 + (SB ? 2 : 4) * HALF);
    if (PRO == FEMASR_PRO_GN_SILU) {
        for (int c = t; c < p.Cin; c += NT) {
            gco[c] = p.pro_a[(size_t)n * p.Cin + c];
            gco[p.Cin + c] = p.pro_b[(size_t)n * p.Cin + c];
        }
    }
    float4 rp[GS];
    auto load_patch = [&](int cc, int g) {          // group g: units [g*GS, min((g+1)*GS, PUNITS))
#pragma unroll
        for (int i = 0; i < GS; ++i)
            if (g * GS + i < PUNITS) rp[i] = ld4(p.in + (size_t)poff[g * GS + i] + (size_t)cc * BK);
    };
    auto store_patch_unit = [&](int buf, int i, int cc) {
        int pix = (t >> 3) + PROWS * i;
        if (PP % PROWS != 0 && i == PUNITS - 1) pix = pix < PP ? pix : PP;     // no branch: keeps the wait counters exact
        float4 v = rp[i % GS];
        if (PRO == FEMASR_PRO_GN_SILU) {
            const float4 ga = *reinterpret_cast<const float4 *>(gco + cc * BK + 4 * kq);
            const float4 gb = *reinterpret_cast<const float4 *>(gco + p.Cin + cc * BK + 4 * kq);
            v.x = fast_silu(__builtin_fmaf(v.x, ga.x, gb.x));
            v.y = fast_silu(__builtin_fmaf(v.y, ga.y, gb.y));
            v.z = fast_silu(__builtin_fmaf(v.z, ga.z, gb.z));
            v.w = fast_silu(__builtin_fmaf(v.w, ga.w, gb.w));
        }
        if (!(pmask & (1u << i))) v = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned h01, l01, h23, l23;
        split_pair(v.x, v.y, h01, l01);
        split_pair(v.z, v.w, h23, l23);
        unsigned short *dst = Ps + (buf * 2) * HALF + pix * PPITCH + 4 * kq;
        *reinterpret_cast<uint2 *>(dst) = make_uint2(h01, h23);
        *reinterpret_cast<uint2 *>(dst + HALF) = make_uint2(l01, l23);
    };

    // accumulators start from the residuals (this path is tolerance-based, so the order of the final additions is free);
    // the bias is added right before the main loop.  The loads are issued as ONE branch-free batch (out-of-range elements
    // read element 0 and are never stored): a per-element `if (in range) v += res[o]` makes the compiler wait for each
    // load before issuing the next, i.e. TM*TN*16 serialised HBM round trips per block (measured: 44 % of the block's
    // lifetime before this change).
    // Output addressing = uniform part (SGPRs) + one per-lane offset: element r of row tile i sits at pixel row
    // 2*(wm*TM+i) + (r>>3), pixel column (r&3) + 8*((r>>2)&1) + 4*(lane>>5) of the 8x16 tile (conv3x3_halo_kernel's scheme).
    const size_t obase = (((size_t)n * p.Ho + oy0) * p.Wo + ox0) * p.Cout + n0;             // uniform
    const unsigned loff4 = 4u * ((unsigned)(4 * (lane >> 5)) * (unsigned)p.Cout + (unsigned)(lane & 31));      // bytes
    auto uoff = [&](int i, int j, int r) -> size_t {            // uniform
        return obase + (size_t)((2 * (wm * TM + i) + (r >> 3)) * p.Wo + (r & 3) + 8 * ((r >> 2) & 1)) * p.Cout + (wn * TN + j) * 32;
    };
    auto ok_u = [&](int i, int j, int r) -> bool {
        return (oy0 + 2 * (wm * TM + i) + (r >> 3)) < p.Ho && (ox0 + (r & 3) + 8 * ((r >> 2) & 1)) < p.Wo && (n0 + (wn * TN + j) * 32) < p.Cout;
    };
    auto ok_l = [&](int i, int j, int r) -> bool {
        return (oy0 + 2 * (wm * TM + i) + (r >> 3)) < p.Ho && (ox0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * (lane >> 5)) < p.Wo &&
               (n0 + (wn * TN + j) * 32 + (lane & 31)) < p.Cout;
    };
    auto ld_res = [&](const float *res, int i, int j, int r) -> float {     // clamped: out-of-range elements read the tile origin
        return ldg_u32(res + (ok_u(i, j, r) ? uoff(i, j, r) : obase), ok_l(i, j, r) ? loff4 : 0u);
    };
    f32x16 acc[TM][TN];
    const float *ra = p.res1 ? p.res1 : p.res2, *rb = (p.res1 && p.res2) ? p.res2 : nullptr;
    if (ra && !DBG16_ON(1)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = ld_res(ra, i, j, r);
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    // split weights, fragment-major: [q][ntile][kstep(2)][hi|lo][lane] x 8 bf16: every wave-level load is one contiguous KiB
    const uint4 *wl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wl[j] = wsplit + ((size_t)wtile(n0, wn * TN + j, p.NT32) * 256 + lane);
    const size_t wstride = (size_t)p.NT32 << 8;     // uint4 per K chunk

    const int ncc = p.Cin / BK;
    if (!DBG16_ON(2)) load_patch(0, 0);
    uint4 bc[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) bc[j][e] = wl[j][e * 64];
    if (PRO == FEMASR_PRO_GN_SILU) __syncthreads();     // gco visible
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g > 0 && !DBG16_ON(2)) load_patch(0, g);
#pragma unroll
        for (int i = 0; i < GS; ++i)
            if (g * GS + i < PUNITS) store_patch_unit(0, g * GS + i, 0);
    }
    __syncthreads();

    int py[TM], px;
    {
        const int m = wm * TM * 32 + (lane & 31);
        px = m & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i) py[i] = (m >> 4) + 2 * i;
    }
    const int koff = 8 * (lane >> 5);                // this lane's k sub-block inside a 16-deep k-step

    // LDS index of this lane's A-fragment pixel for (tap, row tile i).  Without the fused x2 upsample it is ONE per-lane
    // base plus a compile-time constant (folded into the ds_read offset field); with it the halving depends on the lane.
    const int abase = (py[0] * PW + px) * PPITCH;
    auto patch_idx = [&](int tap, int (&idx)[TM]) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (UP2) {
                const int prow = ((py[i] + ky - 1) >> 1) + 1, pcol = ((px + kx - 1) >> 1) + 1;
                idx[i] = (prow * PW + pcol) * PPITCH;
            } else {
                idx[i] = abase + ((2 * i + ky) * PW + kx) * PPITCH;
            }
        }
    };

    // Main loop.  Everything inside is UNCONDITIONAL straight-line code (9 taps unrolled; the last channel block
    // re-stages itself into the idle LDS buffer and re-reads the last weight chunk) so that the compiler knows exactly
    // how many loads are in flight and emits exact s_waitcnt values; a conditional load costs a vmcnt(0) per k-step.
    const int nq = ncc * 9;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            const float bv = col < p.Cout ? p.bias[col] : 0.f;
            if (rb) {           // second residual (rare: last ResBlock of an up block): one batch per 32x32 tile
                float tmp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) tmp[r] = ld_res(rb, i, j, r);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += tmp[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
        }
    TT_STAMP_ALWAYS(11)
    for (int cc = 0; cc < ncc; ++cc) {
        const unsigned short *Pb = Ps + (SB ? 0 : ((cc & 1) * 2) * HALF) + koff;
        const int ccn = cc + 1 < ncc ? cc + 1 : cc;
        const int nbuf = SB ? 0 : (cc + 1) & 1;
        if constexpr (ROT) {
            uint4 a_hi[TM], a_lo[TM];
            int aidx[TM], nidx[TM];
            patch_idx(0, aidx);
#pragma unroll
            for (int i = 0; i < TM; ++i) a_hi[i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap > 0) TT_STAMP(tap - 1)
                const int q1 = cc * 9 + tap + 1;
                const size_t qn = (size_t)(q1 < nq ? q1 : nq - 1);
                if (tap == 0) load_patch(ccn, 0);
                patch_idx(tap < 8 ? tap + 1 : 8, nidx);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a_lo[i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + 16 * s + HALF);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int term = 0; term < 2; ++term)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi[i]), as_bf16x8(bc[j][2 * s + term]),
                                                                                    acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // the hi registers are free: fetch the next k-step's (or next tap's) hi fragments under the lo term
                    if (s == 0) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) a_hi[i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + 16);
                    } else if (tap < 8) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) a_hi[i] = *reinterpret_cast<const uint4 *>(Pb + nidx[i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo[i]), as_bf16x8(bc[j][2 * s]), acc[i][j],
                                                                                0, 0, 0);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bc[j][2 * s] = (wl[j] + qn * wstride)[(2 * s) * 64];
                        bc[j][2 * s + 1] = (wl[j] + qn * wstride)[(2 * s + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) aidx[i] = nidx[i];
                // next channel block's patch in three groups (registers for one group only): group g is loaded at tap
                // 3g (right after the previous group's store) and normalised / split / stored at tap 3g + 2
                if (tap % 3 == 2) {
                    const int g = tap / 3;
#pragma unroll
                    for (int i = 0; i < GS; ++i)
                        if (g * GS + i < PUNITS) store_patch_unit(nbuf, g * GS + i, ccn);
                    if (g + 1 < NG && (g + 1) * GS < PUNITS) load_patch(ccn, g + 1);
                }
            }
        } else {
            uint4 a_hi[2][TM], a_lo[2][TM];
            int aidx[TM];
            patch_idx(0, aidx);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a_hi[0][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i]);
                a_lo[0][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + HALF);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap > 0) TT_STAMP(tap - 1)
                const int q1 = cc * 9 + tap + 1;
                const size_t qn = (size_t)(q1 < nq ? q1 : nq - 1);
                if (tap == 0) load_patch(ccn, 0);
                int nidx[TM];
                patch_idx(tap < 8 ? tap + 1 : 8, nidx);
                // A fragments are fetched one k-step ahead (s=1 while s=0 multiplies, the next tap's s=0 while s=1
                // multiplies), pinned with sched_barrier so the LDS latency hides behind MFMAs
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s == 0) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            a_hi[1][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + 16);
                            a_lo[1][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + 16 + HALF);
                        }
                    } else if (tap < 8) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            a_hi[0][i] = *reinterpret_cast<const uint4 *>(Pb + nidx[i]);
                            a_lo[0][i] = *reinterpret_cast<const uint4 *>(Pb + nidx[i] + HALF);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // term-major order: consecutive MFMAs hit DIFFERENT accumulator tiles, so the dependent-accumulator
                    // latency of the 32x32x16 bf16 MFMA (longer than its issue interval) is hidden
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                const bf16x8 a = as_bf16x8(term == 2 ? a_lo[s][i] : a_hi[s][i]);
                                const bf16x8 b = as_bf16x8(term == 1 ? bc[j][2 * s + 1] : bc[j][2 * s]);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
                            }
                    // this k-step's weight registers are free now: refill them with the NEXT tap's fragments (one tap of
                    // MFMAs ahead of their use), no register copies
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bc[j][2 * s] = (wl[j] + qn * wstride)[(2 * s) * 64];
                        bc[j][2 * s + 1] = (wl[j] + qn * wstride)[(2 * s + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) aidx[i] = nidx[i];
                // next channel block's patch: loaded at tap 0, one unit normalised / split / stored per tap over the LAST
                // PUNITS taps (bf16 taps are ~5x shorter than fp32 ones: the HBM latency needs the distance)
                if (!SB && tap >= 9 - PUNITS) store_patch_unit(nbuf, tap - (9 - PUNITS), ccn);
            }
            if (SB && cc + 1 < ncc) {       // everyone is done reading the (only) buffer: re-fill it with the next channel block
                __syncthreads();
#pragma unroll
                for (int i = 0; i < PUNITS; ++i) store_patch_unit(0, i, ccn);
            }
        }
        TT_STAMP(8)
        __syncthreads();
        TT_STAMP(9)
    }
#if defined(FEMASR_TAPTIME) && FEMASR_TAPTIME < 2
    TT_STAMP_ALWAYS(0)
#endif

    // ---- epilogue.  Stores are ISSUE-bound (one dword store per accumulator register = 256 B per wave instruction), so
    // each 32x32 tile is transposed through a per-wave LDS scratch (pitch 36 floats: 16-byte rows, conflict-free both
    // ways) and written with dwordx4 stores: lane l holds channels 4(l&7)..+3 of pixel rows (l>>3) + 8k, k = 0..3, i.e.
    // 8 full 128-byte rows per instruction and 4x fewer store instructions.  Needs Cout % 4 == 0 (all layers but the
    // 3-channel out_conv, which keeps the scalar stores).
    float colsum[TM][TN], colsq[TM][TN];
    const bool full = (oy0 + 8 <= p.Ho) && (ox0 + TW <= p.Wo) && (n0 + BN <= p.Cout);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float ps = 0.f, pss = 0.f;      // this lane's share of the GroupNorm moments of the OUTPUT (its column)
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ps += acc[i][j][r];
                    pss = __builtin_fmaf(acc[i][j][r], acc[i][j][r], pss);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ok_l(i, j, r)) {
                        ps += acc[i][j][r];
                        pss = __builtin_fmaf(acc[i][j][r], acc[i][j][r], pss);
                    }
            }
            colsum[i][j] = ps;
            colsq[i][j] = pss;
        }
    if ((p.Cout & 3) == 0) {
        float *T = reinterpret_cast<float *>(smem_u16) + 2048 + wave * (32 * 36);      // 8 KB in: clear of the GN `red` area
        const int trow = lane >> 3, tq = lane & 7;
        const unsigned lvec4 = 4u * ((unsigned)trow * (unsigned)p.Cout + 4u * (unsigned)tq);      // bytes
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[i][j][r];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 v = *reinterpret_cast<const float4 *>(T + (trow + 8 * k) * 36 + 4 * tq);
                    const int prow = 2 * (wm * TM + i) + (k >> 1), pcol0 = 8 * (k & 1);               // uniform
                    const bool okv = full || ((oy0 + prow) < p.Ho && (ox0 + pcol0 + trow) < p.Wo && (n0 + (wn * TN + j) * 32 + 4 * tq) < p.Cout);
                    if (okv) {
                        float *ub = p.out + obase + (size_t)(prow * p.Wo + pcol0) * p.Cout + (wn * TN + j) * 32;
                        const unsigned long long a = uniform_u64(reinterpret_cast<unsigned long long>(ub));
                        *reinterpret_cast<__attribute__((address_space(1))) f32x4 *>(a + lvec4) = f32x4{v.x, v.y, v.z, v.w};
                    }
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ok_l(i, j, r)) stg_u32(p.out + uoff(i, j, r), loff4, acc[i][j][r]);
    }

    // Optional fused GroupNorm moments of the output (consumed by the NEXT conv's GN prologue): per (tile, group)
    // partial sums, reduced lane -> group (xor shuffles over the cg lanes of a group, then the two row halves) -> waves
    // (LDS) and written as doubles to stats_part[((n*tiles + tile)*32 + g)*2]; a fixed order, so runs are reproducible.
#ifdef FEMASR_TAPTIME
    TT_STAMP_ALWAYS(12)
    tt[10] = tprev - tstart;
    if (lane == 0) {
        const unsigned slot = (blockIdx.x * (WM * WN) + wave) & 65535u;
        for (int i = 0; i < 13; ++i) g_taptime[slot * 16 + i] += tt[i];
    }
#endif
    if (stats_part) {
        const int cg = p.Cout >> 5;                       // channels per group (32 groups): 8 / 4 / 2
        double *red = reinterpret_cast<double *>(smem_u16);   // [WM][BN][2] (patch buffers are dead after the last barrier)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            double s_ = 0.0, q_ = 0.0;      // cross-lane / cross-wave part in fp64 (per-lane partials are <= 32 fp32 terms)
#pragma unroll
            for (int i = 0; i < TM; ++i) { s_ += (double)colsum[i][j]; q_ += (double)colsq[i][j]; }
            for (int sft = 1; sft < cg; sft <<= 1) {
                s_ += __shfl_xor(s_, sft, 64);
                q_ += __shfl_xor(q_, sft, 64);
            }
            s_ += __shfl_xor(s_, 32, 64);
            q_ += __shfl_xor(q_, 32, 64);
            if (lane < 32 && (lane % cg) == 0) {
                const int gl = ((wn * TN + j) * 32 + lane) / cg;        // group index inside this block's BN columns
                red[(wm * BN + gl) * 2] = s_;
                red[(wm * BN + gl) * 2 + 1] = q_;
            }
        }
        __syncthreads();
        const int ngl = BN / cg;                                         // groups covered by this block
        if (t < ngl && n0 + t * cg < p.Cout) {
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) {
                S += red[(w2 * BN + t) * 2];
                Q += red[(w2 * BN + t) * 2 + 1];
            }
            const int g = n0 / cg + t;
            const size_t tile_id = (size_t)n * p.tilesX * p.tilesY + (size_t)ty * p.tilesX + tx;
            stats_part[(tile_id * 32 + g) * 2] = S;
            stats_part[(tile_id * 32 + g) * 2 + 1] = Q;
        }
    }
}

// OIHW fp32 -> split bf16 fragment-major: out ushort index = (((((q*NT32 + ntile)*2 + s)*2 + h)*64 + lane)*8 + e)
//   k = q*32 + 16 s + 8 (lane>>5) + e  in the blocked K order, n = ntile*32 + (lane&31), h: 0 = hi, 1 = lo
__device__ __forceinline__ unsigned short bf16_rne(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ void repack_oihw_bf16x3_kernel(const float *__restrict__ in, int O, int I, int kh, int kw,
                                          unsigned short *__restrict__ out, size_t total)
{
    const int K = I * kh * kw, NT32 = (O + 31) / 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), h = (int)((i >> 9) & 1), s = (int)((i >> 10) & 1);
        const size_t rest = i >> 11;
        const int ntile = (int)(rest % NT32), q = (int)(rest / NT32);
        const int k = q * 32 + 16 * s + 8 * (lane >> 5) + e, o = ntile * 32 + (lane & 31);
        float v = 0.f;
        if (k < K && o < O) {
            const int cl = k % 32;
            int r = k / 32;
            const int x = r % kw;
            r /= kw;
            const int y = r % kh;
            const int ci = (r / kh) * 32 + cl;
            v = in[(((size_t)o * I + ci) * kh + y) * kw + x];
        }
        const unsigned short hi = bf16_rne(v);
        const float rem = v - __uint_as_float((unsigned)hi << 16);
        out[i] = h == 0 ? hi : bf16_rne(rem);
    }
}

template <int BN, bool UP2>
constexpr size_t bf16_lds_bytes() { return (size_t)((BN <= 64 && !UP2) ? 1 : 2) * 2 * ((UP2 ? 60 : 180) + 1) * PPITCH * sizeof(unsigned short); }   // + 2*Cin floats (GN)

struct Variant16 {
    const char *name;
    int bn, threads;
    void (*kern)(const ConvParams, const uint4 *, double *);
    size_t lds;
    unsigned long long attr_devs;       // bit d: MaxDynamicSharedMemorySize set on device d (the attribute is per device)
};
#define FEMASR_H16(BN, WM, WN, PRO, UP2)                                                        \
    { "conv3x3_halo_bf16x3<8x16x" #BN "," #PRO ",up2=" #UP2 ",waves=" #WM "x" #WN ">", BN, WM * WN * 64,   \
      conv3x3_halo_bf16x3_kernel<BN, WM, WN, PRO, UP2>, bf16_lds_bytes<BN, UP2>(), 0ull }

Variant16 g_v16[] = {
    FEMASR_H16(128, 2, 2, FEMASR_PRO_NONE, false),     // 0   (4 waves: 64 px x 64 ch per wave)
    FEMASR_H16(128, 2, 2, FEMASR_PRO_GN_SILU, false),  // 1
    FEMASR_H16(128, 2, 2, FEMASR_PRO_NONE, true),      // 2
    FEMASR_H16(64, 4, 2, FEMASR_PRO_NONE, false),      // 3   (8 waves: 32 px x 32 ch per wave)
    FEMASR_H16(64, 4, 2, FEMASR_PRO_GN_SILU, false),   // 4
    FEMASR_H16(64, 4, 1, FEMASR_PRO_NONE, true),       // 5   (4 waves)
    FEMASR_H16(32, 4, 1, FEMASR_PRO_NONE, false),      // 6
    FEMASR_H16(32, 4, 1, FEMASR_PRO_GN_SILU, false),   // 7
    FEMASR_H16(32, 4, 1, FEMASR_PRO_NONE, true),       // 8
    FEMASR_H16(256, 1, 4, FEMASR_PRO_NONE, false),     // 9   (4 waves: 128 px x 64 ch per wave, one column block for Cout = 256)
    FEMASR_H16(256, 1, 4, FEMASR_PRO_GN_SILU, false),  // 10
    FEMASR_H16(256, 1, 4, FEMASR_PRO_NONE, true),      // 11
    FEMASR_H16(64, 4, 1, FEMASR_PRO_NONE, false),      // 12  (4 waves: 32 px x 64 ch per wave)
    FEMASR_H16(64, 4, 1, FEMASR_PRO_GN_SILU, false),   // 13
    FEMASR_H16(64, 4, 1, FEMASR_PRO_NONE, true),       // 14
    FEMASR_H16(64, 2, 2, FEMASR_PRO_NONE, false),      // 15  (4 waves: 64 px x 32 ch per wave)
    FEMASR_H16(64, 2, 2, FEMASR_PRO_GN_SILU, false),   // 16
    FEMASR_H16(64, 2, 2, FEMASR_PRO_NONE, true),       // 17
};
constexpr int kNum16 = sizeof(g_v16) / sizeof(g_v16[0]);

}  // namespace

int femasr_conv_bf16x3_variant_count() { return kNum16; }
const char *femasr_conv_bf16x3_variant_name(int v) { return (v >= 0 && v < kNum16) ? g_v16[v].name : "?"; }

bool femasr_conv_bf16x3_eligible(const femasr_conv_args *a) { return a->w_bf16x3 && femasr_conv_bf16x3_shape_ok(a); }

bool femasr_conv_bf16x3_shape_ok(const femasr_conv_args *a)
{
    return a->ksz == 3 && a->stride == 1 && a->pad == 1 && (a->Cin % BK) == 0 &&
           a->prologue != FEMASR_PRO_LN && a->act == FEMASR_ACT_NONE && !(a->up2 && a->prologue != FEMASR_PRO_NONE) &&
           a->Cin <= 1024 && (size_t)a->B * a->H * a->W * a->Cin < ((size_t)1 << 31) &&
           (size_t)a->B * a->H * a->W * (a->up2 ? 4 : 1) * a->Cout < ((size_t)1 << 31);
}

int femasr_conv_bf16x3_launch(hipStream_t s, const femasr_conv_args *a, int *variant_out, double *flops_out)
{
    FEMASR_REQUIRE(a && a->in && a->bias && a->out && femasr_conv_bf16x3_eligible(a), "conv bf16x3: not eligible");
    const int Hv = a->up2 ? 2 * a->H : a->H, Wv = a->up2 ? 2 * a->W : a->W;
    FEMASR_REQUIRE(Hv == a->Ho && Wv == a->Wo, "conv bf16x3: Ho/Wo mismatch");
    if (a->prologue == FEMASR_PRO_GN_SILU) FEMASR_REQUIRE(a->pro_a && a->pro_b, "conv bf16x3: GN prologue needs a,b");
    ConvParams p{};
    p.in = a->in; p.bias = a->bias; p.pro_a = a->pro_a; p.pro_b = a->pro_b; p.res1 = a->res1; p.res2 = a->res2; p.out = a->out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.ksz = 3; p.stride = 1; p.pad = 1; p.up2 = a->up2;
    p.Ho = Hv; p.Wo = Wv; p.NT32 = (a->Cout + 31) / 32;
    int cls = a->Cout > 128 ? 3 : (a->Cout > 64 ? 0 : (a->Cout > 32 ? 1 : 2));
#ifdef FEMASR_TAPTIME
    if (getenv("FEMASR_BF16_CLS")) cls = atoi(getenv("FEMASR_BF16_CLS"));      // debug build only: force a tile class
#endif
    // Cout 33..64 uses the 4-wave 64 px x 32 ch tiling (rows 15..17: +2.5 % / +9 % fused-x2 over the 8-wave rows 3..5)
    const int vi = (cls == 1 ? 15 : cls * 3) + (a->up2 ? 2 : a->prologue);
    Variant16 &v = g_v16[vi];
    p.tilesX = (p.Wo + 15) / 16;
    p.tilesY = (p.Ho + 7) / 8;
    p.MB = a->B * p.tilesX * p.tilesY;
    p.NB = (a->Cout + v.bn - 1) / v.bn;
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((__atomic_load_n(&v.attr_devs, __ATOMIC_ACQUIRE) >> dev) & 1ull)) {      // (idempotent: a race only repeats the call)
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds + 40 * 1024));
        if (dev >= 0 && dev < 64) __atomic_fetch_or(&v.attr_devs, 1ull << dev, __ATOMIC_RELEASE);
    }
    size_t lds = v.lds + (a->prologue == FEMASR_PRO_GN_SILU ? (size_t)2 * a->Cin * sizeof(float) : 0);
    const size_t epi = 8192 + (size_t)(v.threads / 64) * 32 * 36 * sizeof(float);       // epilogue transpose scratch
    if (lds < epi) lds = epi;
    FEMASR_REQUIRE(!a->gn_part || (a->Cout % 32 == 0 && (a->Cout / 32) <= 8 && ((a->Cout / 32) & (a->Cout / 32 - 1)) == 0),
                   "conv bf16x3: fused GN moments need Cout = 32 * {1, 2, 4, 8} (32 groups, power-of-two channels per group)");
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(p.MB * p.NB)), dim3((unsigned)v.threads), lds, s, p, (const uint4 *)a->w_bf16x3,
                       (double *)a->gn_part);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (variant_out) *variant_out = vi;
    if (flops_out) *flops_out = 2.0 * (double)a->B * p.Ho * p.Wo * (double)a->Cout * 9.0 * a->Cin;
    return FEMASR_OK;
}

#ifdef FEMASR_TAPTIME
extern "C" int femasr_debug_set_flags16(int flags)
{
    hipMemcpyToSymbol(HIP_SYMBOL(g_dbg16_flags), &flags, sizeof(int));
    return 0;
}
extern "C" int femasr_debug_taptime(unsigned long long *out16, int reset)
{
    static unsigned long long host[16 * 65536];
    if (out16) {
        hipMemcpyFromSymbol(host, HIP_SYMBOL(g_taptime), sizeof(host));
        for (int i = 0; i < 16; ++i) out16[i] = 0;
        for (int w = 0; w < 65536; ++w)
            for (int i = 0; i < 16; ++i) out16[i] += host[w * 16 + i];
    }
    if (reset) {
        void *d = nullptr;
        hipGetSymbolAddress(&d, HIP_SYMBOL(g_taptime));
        hipMemset(d, 0, sizeof(host));
    }
    return 0;
}
#endif

extern "C" {

size_t femasr_packed_weight_bf16x3_bytes(int O, int I, int kh, int kw)
{
    if (O <= 0 || I <= 0 || kh <= 0 || kw <= 0 || (I % 32) != 0) return 0;
    const size_t K = (size_t)I * kh * kw;
    return (K / 32) * (size_t)((O + 31) / 32) * 2048 * sizeof(unsigned short);
}

int femasr_repack_oihw_bf16x3(void *stream, const float *in, int O, int I, int kh, int kw, void *out)
{
    FEMASR_REQUIRE(in && out && O > 0 && I > 0 && kh > 0 && kw > 0 && (I % 32) == 0, "repack bf16x3: needs I %% 32 == 0");
    const size_t total = femasr_packed_weight_bf16x3_bytes(O, I, kh, kw) / sizeof(unsigned short);
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(repack_oihw_bf16x3_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, O, I, kh, kw,
                       (unsigned short *)out, total);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

}  // extern "C"
