// kernels_conv.hip — NHWC convolution / linear / VQ-distance kernels on the gfx950 fp32 matrix cores.
//
// Two kernel families cover every dense contraction of the hot path:
//   conv3x3_halo  3x3 stride-1 pad-1 convs (753 of the 964 GFLOP per tile), optional fused nearest-x2
//                 (nn.Upsample, femasr_arch.py:172,202) and fused GroupNorm-apply + SiLU (fema_utils.py:72-79)
//   conv_igemm    the remaining general convs as an implicit GEMM: k4 in_conv, stride-2 convs, odd shapes
//                 (1x1 convs / nn.Linear / the VQ distance matrix live in kernels_gemm.hip)
//   both          bias, exact-erf GELU, up to two residual adds on store (fema_utils.py:82-83;
//                 network_swinir.py:276-277,482; femasr_arch.py:361-362)
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32, and per output element ONE fmaf chain in the oracle's K
// order  k = ((ci/32)*ksz*ksz + ky*ksz + kx)*32 + ci%32  (plain (ky,kx,ci) when Cin % 32 != 0), so results are
// bit-identical to oracle/femasr_oracle.c.  That fixes the schedule: no split-K, one accumulator per output.
//
// Operands:
//   A (activations)  staged global -> registers -> (normalise / activate) -> LDS [pixel][33] (the pad makes both
//                    the MFMA A-fragment read  A[i=lane&31][k=lane>>5]  and the scattered stores conflict-free).
//                    conv3x3_halo stages ONE (8+2)x(16+2) halo patch per 32-channel block and sweeps all 9 taps
//                    over it (6.4x fewer loads / activations / LDS stores than im2col staging).
//   B (weights)      never touch LDS: femasr_repack_oihw stores them FRAGMENT-MAJOR,
//                        [chunk q = k/32][n-tile = n/32][lane = (k&1)*32 + n%32][kk = (k%32)/2]      (zero padded)
//                    so the 16 B-fragment values a lane needs for one 32-deep K chunk of one 32-column tile are 64
//                    contiguous bytes and a wave reads 4 KiB contiguous with four global_load_dwordx4, straight
//                    into the registers the MFMAs consume (weights are L2 / L1 resident: <= 9.4 MB per layer).
//                    No weight staging, no B bank traffic, and the halo kernel needs ONE barrier per channel
//                    block instead of one per tap.
// Blocks: 128 output pixels x BN channels, BK = 32.  Cout > 32: 256 threads = 4 waves of 64 x 64 (or 64 px x 32 ch)
//   outputs, 2-3 blocks per CU; Cout <= 32 (out_conv) and the odd shapes: 512 threads = 8 waves of 32 x 64 / 32 x 32.
// Main loops are UNCONDITIONAL straight-line code (last chunk re-stages itself into the idle buffer) so that the
//   compiler's s_waitcnt values are exact; epilogues load residuals in branch-free per-tile batches.
// Grid: 1-D, n-blocks fastest, bijective XCD remap (block b runs on XCD b % 8) so tiles sharing activations /
//   halo rows share an L2.
#include "conv_common.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include "detmath.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// C/D layout of a 32x32 MFMA tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).

// =================================================================================================================
// conv_igemm: implicit GEMM, M = B*Ho*Wo pixels, N = Cout, K chunks of 32
// =================================================================================================================
template <int BM, int BN, int WM, int WN, int PRO, bool CINVEC>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 4 : 2) void conv_igemm_kernel(const ConvParams p)
{
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int RSTEP = NT / 8;                // A rows covered per unit round
    constexpr int AROWS = BM / RSTEP;            // A float4 units per thread
    static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && AROWS >= 1, "tile config");
    static_assert(CINVEC || PRO == FEMASR_PRO_NONE, "generic-Cin path has no prologue");
    static_assert(PRO != FEMASR_PRO_LN, "LayerNorm is a separate pass (femasr_layernorm)");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [2][BM][ALD]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // provably wave-uniform -> SGPR arithmetic
    const int wm = wave / WN, wn = wave % WN;
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB, mb = L / p.NB;
    const int m0 = mb * BM, n0 = nb * BN;

    // ---- per-thread A rows: (mrow + RSTEP j, k-quad kq)
    const int kq = t & 7, mrow = t >> 3;
    int rn[AROWS], riy[AROWS], rix[AROWS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
        const int r = m0 + mrow + RSTEP * j;
        if (r < p.M) {
            const int n = r / HoWo, rem = r - n * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            rn[j] = n;
            riy[j] = oy * p.stride - p.pad;
            rix[j] = ox * p.stride - p.pad;
        } else {
            rn[j] = 0;
            riy[j] = -(1 << 28);
            rix[j] = -(1 << 28);
        }
    }
    const int Hv = p.up2 ? 2 * p.H : p.H, Wv = p.up2 ? 2 * p.W : p.W;

    float4 ra[AROWS], rga[AROWS], rgb[AROWS];
    unsigned amask = 0;

    // ---- global -> register staging of the A part of K-chunk c (branch-free: out-of-image taps / tail rows read
    //      element 0 and are zeroed at store time, so the loads stay in flight across the MFMA steps)
    auto load_chunk = [&](int c) {
        amask = 0;
        if (CINVEC) {
            const int cc = c / p.taps, tap = c - cc * p.taps, c0 = cc * BK + 4 * kq;
            const int ky = tap / p.ksz, kx = tap - ky * p.ksz;
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                const int iy = riy[j] + ky, ix = rix[j] + kx;
                const bool ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv);
                const int sy = p.up2 ? (iy >> 1) : iy, sx = p.up2 ? (ix >> 1) : ix;
                const size_t off = ok ? ((((size_t)rn[j] * p.H + sy) * p.W + sx) * p.Cin + c0) : (size_t)0;
                ra[j] = ld4(p.in + off);
                if (PRO == FEMASR_PRO_GN_SILU) {
                    rga[j] = ld4(p.pro_a + (size_t)rn[j] * p.Cin + c0);
                    rgb[j] = ld4(p.pro_b + (size_t)rn[j] * p.Cin + c0);
                }
                amask |= (ok ? 1u : 0u) << j;
            }
        } else {
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = c * BK + 4 * kq + e;
                    v[e] = 0.f;
                    if (k < p.K) {
                        const int tap = k / p.Cin, ci = k - tap * p.Cin;
                        const int ky = tap / p.ksz, kx = tap - ky * p.ksz;
                        const int iy = riy[j] + ky, ix = rix[j] + kx;
                        if ((iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv)) {
                            const int sy = p.up2 ? (iy >> 1) : iy, sx = p.up2 ? (ix >> 1) : ix;
                            v[e] = p.in[(((size_t)rn[j] * p.H + sy) * p.W + sx) * p.Cin + ci];
                        }
                    }
                }
                ra[j] = make_float4(v[0], v[1], v[2], v[3]);
                amask |= 1u << j;
            }
        }
    };

    // ---- registers -> LDS (with the fused normalisation / activation prologue)
    auto store_chunk = [&](int buf) {
        float *Ab = As + buf * BM * ALD;
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            // opaque use: pins the prologue arithmetic HERE (after the chunk's MFMAs); without it the optimizer hoists part
            // of it to just behind the loads and the wave stalls on the HBM latency at the top of every chunk
            asm volatile("" : "+v"(ra[j].x), "+v"(ra[j].y), "+v"(ra[j].z), "+v"(ra[j].w));
            float4 v = ra[j];
            if (PRO == FEMASR_PRO_GN_SILU) {
                v.x = det_silu(__builtin_fmaf(v.x, rga[j].x, rgb[j].x));
                v.y = det_silu(__builtin_fmaf(v.y, rga[j].y, rgb[j].y));
                v.z = det_silu(__builtin_fmaf(v.z, rga[j].z, rgb[j].z));
                v.w = det_silu(__builtin_fmaf(v.w, rga[j].w, rgb[j].w));
            }
            if (!(amask & (1u << j))) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies AFTER the activation
            float *dst = Ab + (mrow + RSTEP * j) * ALD + 4 * kq;
            dst[0] = v.x;
            dst[1] = v.y;
            dst[2] = v.z;
            dst[3] = v.w;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B fragments: lane-contiguous 16 floats per (chunk, 32-column tile); chunk stride = NT32 * 1024 floats.
    // 1x1 layers with Cin % 32 == 0 that are not plain GEMMs (padded / strided / GN prologue) still carry the GEMM weight
    // layout [chunk][ntile][j][lane][t] and its k order (ORC_KPERM): 4 floats per lane and 8-channel group, groups 1 KiB apart.
    const bool kp = p.kperm != 0;             // uniform
    const int wgs = kp ? 256 : 4;             // floats between consecutive 4-step weight groups of a lane
    const float *wl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const size_t tile = (size_t)wtile(n0, wn * TN + j, p.NT32);
        wl[j] = kp ? p.w + tile * 1024 + lane * 4 : p.w + ((tile * 64 + lane) << 4);
    }
    const size_t wstride = (size_t)p.NT32 << 10;

    load_chunk(0);
    float4 bc[TN], bn[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bc[j] = ld4(wl[j]);
    store_chunk(0);
    __syncthreads();

    const int arow = (wm * TM * 32 + (lane & 31)) * ALD + (kp ? 4 : 1) * (lane >> 5);
    auto kofs = [&](int kk) -> int { return kp ? 8 * (kk >> 2) + (kk & 3) : 2 * kk; };       // A column of MFMA step kk (lanes 0-31)

    // The loop body is unconditional straight-line code: the last chunk re-loads itself and re-stages it into the idle
    // LDS buffer, so the compiler knows exactly how many loads are in flight at every wait (a conditional load forces a
    // conservative s_waitcnt vmcnt(0) at the next use of ANY loaded value).
    for (int c = 0; c < p.nchunks; ++c) {
        const int buf = c & 1;
        const int cn = (c + 1) < p.nchunks ? c + 1 : c;
        const float *Ab = As + buf * BM * ALD + arow;
        float af[2][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[i * 32 * ALD];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1, g = kk >> 2, e = kk & 3;
            if (e == 0) {       // prefetch the next 4 k-pairs of weight fragments (next chunk's first 4 at the end)
                if (g < 3) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) bn[j] = ld4(wl[j] + (size_t)c * wstride + wgs * (g + 1));
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) bn[j] = ld4(wl[j] + (size_t)cn * wstride);
                }
                // next chunk's A rows: issued right AFTER a weight prefetch.  Loads complete in order, so the first
                // younger weight load that is waited for (two prefetch groups = 16 MFMAs later) also waits for these
                // HBM loads; issued before the prefetch they would only get 8 MFMAs of cover.
                if (kk == 0) load_chunk(cn);
            }
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = Ab[i * 32 * ALD + kofs(kk + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], f4get(bc[j], e), acc[i][j], 0, 0, 0);
            if (e == 3) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bc[j] = bn[j];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        store_chunk(buf ^ 1);
        __syncthreads();
    }

    {
        // store: out = act(acc + bias) + res1 + res2, in that order (the bit-exact contract).  Address = uniform (tile,
        // wave, i, j, r) part in SGPRs + one per-lane offset.  On full tiles the residuals are loaded as branch-free batches
        // of one 32x32 tile (16 values per lane), ONE TILE AHEAD of the tile being stored: a per-element `load; add; store`
        // chain costs one HBM round trip per element (16*TM*TN per block - more than the K = 256 main loop of the Swin
        // linears).
        const unsigned loff = (unsigned)((lane >> 5) * 4) * (unsigned)p.Cout + (unsigned)(lane & 31);
        const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.Cout);
        const float *ra = p.res1 ? p.res1 : p.res2, *rb = (p.res1 && p.res2) ? p.res2 : nullptr;
        auto uoff = [&](int i, int j, int r) -> size_t {       // uniform
            return (size_t)(m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * p.Cout + (n0 + (wn * TN + j) * 32);
        };
        // edge tiles: the same code with clamped addresses (uniform part -> this tile's first element, lane part -> 0) for
        // the loads and masked stores, so the address stays "SGPR base + one VGPR" everywhere
        auto epilogue = [&](auto nres_c, auto full_c) {
            constexpr int NRES = decltype(nres_c)::value;
            constexpr bool FULL = decltype(full_c)::value;
            constexpr int DEPTH = NRES == 1 ? 2 : 1;     // one residual: loads run a tile ahead; two: per-tile batches
            float rbuf[DEPTH][NRES > 0 ? NRES : 1][16];
            auto ok_u = [&](int i, int j, int r) -> bool {    // uniform: row / column-tile start inside the matrix
                return FULL || ((m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) < p.M && (n0 + (wn * TN + j) * 32) < p.Cout);
            };
            auto ok_l = [&](int i, int j, int r) -> bool {
                return FULL || ((m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) < p.M &&
                                (n0 + (wn * TN + j) * 32 + (lane & 31)) < p.Cout);
            };
            auto issue = [&](int tl, int slot) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = tl / TN, j = tl % TN;
                    const size_t ou = ok_u(i, j, r) ? uoff(i, j, r) : (size_t)m0 * p.Cout + n0;
                    const unsigned lo = ok_l(i, j, r) ? 4u * loff : 0u;
                    if (NRES >= 1) rbuf[slot][0][r] = ldg_u32(ra + ou, lo);
                    if (NRES >= 2) rbuf[slot][NRES >= 2 ? 1 : 0][r] = ldg_u32(rb + ou, lo);
                }
            };
            if (NRES > 0) issue(0, 0);
#pragma unroll
            for (int tl = 0; tl < TM * TN; ++tl) {
                const int i = tl / TN, j = tl % TN;
                if (DEPTH == 2 && tl + 1 < TM * TN) issue(tl + 1, (tl + 1) & 1);
                const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
                const float bv = (FULL || col < p.Cout) ? p.bias[FULL ? col : (col < p.Cout ? col : 0)] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv;
                    if (p.act == FEMASR_ACT_GELU) v = det_gelu(v);
                    if (NRES >= 1) v = v + rbuf[tl % DEPTH][0][r];
                    if (NRES >= 2) v = v + rbuf[tl % DEPTH][NRES >= 2 ? 1 : 0][r];
                    if (ok_l(i, j, r)) stg_u32(p.out + uoff(i, j, r), 4u * loff, v);
                }
                if (NRES > 0 && DEPTH == 1 && tl + 1 < TM * TN) issue(tl + 1, 0);
            }
        };
        // Full tiles with Cout % 4 == 0 (every nn.Linear): stores are ISSUE-bound (a dword store per accumulator register
        // moves 256 B per wave instruction; the K = 256 linears write 64 KB per 16k MFMA-cycles of work), so each 32x32 tile
        // is transposed through a per-wave LDS scratch and bias / GELU / residuals / store run on float4 rows: lane l owns
        // columns 4(l&7)..+3 of tile rows (l>>3) + 8k.  Same per-element arithmetic and order, 4x fewer memory instructions.
        constexpr bool VEC_FITS = (WM * WN * TSCRATCH) <= (2 * BM * ALD);
        auto epilogue_vec = [&](auto nres_c) {
            constexpr int NRES = decltype(nres_c)::value;
            constexpr int DEPTH = NRES == 1 ? 2 : 1;
            float *T = smem + wave * TSCRATCH;           // the A buffers are dead after the loop's last barrier
            const int trow = lane >> 3, tq = lane & 7;
            const unsigned lvec = 4u * ((unsigned)trow * (unsigned)p.Cout + 4u * (unsigned)tq);     // bytes
            auto urow = [&](int i, int j, int k) -> size_t {       // uniform: tile row 8k, tile column 0
                return (size_t)(m0 + (wm * TM + i) * 32 + 8 * k) * p.Cout + (n0 + (wn * TN + j) * 32);
            };
            f32x4_t rbuf[DEPTH][NRES > 0 ? NRES : 1][4];
            auto issue = [&](int tl, int slot) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (NRES >= 1) rbuf[slot][0][k] = ldg4_u32(ra + urow(tl / TN, tl % TN, k), lvec);
                    if (NRES >= 2) rbuf[slot][NRES >= 2 ? 1 : 0][k] = ldg4_u32(rb + urow(tl / TN, tl % TN, k), lvec);
                }
            };
            if (NRES > 0) issue(0, 0);
#pragma unroll
            for (int tl = 0; tl < TM * TN; ++tl) {
                const int i = tl / TN, j = tl % TN;
                if (DEPTH == 2 && tl + 1 < TM * TN) issue(tl + 1, (tl + 1) & 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TPITCH + (lane & 31)] = acc[i][j][r];
                const f32x4_t b4 = ldg4_u32(p.bias + n0 + (wn * TN + j) * 32, 16u * (unsigned)tq);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(T + (trow + 8 * k) * TPITCH + 4 * tq);
                    float v[4] = {a4.x + b4[0], a4.y + b4[1], a4.z + b4[2], a4.w + b4[3]};
                    if (p.act == FEMASR_ACT_GELU) {         // two at a time on the packed fp32 ALU (bit-identical per element)
                        const det_f32x2 g0 = det_gelu2(det_f32x2{v[0], v[1]}), g1 = det_gelu2(det_f32x2{v[2], v[3]});
                        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (NRES >= 1) v[e] = v[e] + rbuf[tl % DEPTH][0][k][e];
                        if (NRES >= 2) v[e] = v[e] + rbuf[tl % DEPTH][NRES >= 2 ? 1 : 0][k][e];
                    }
                    stg4_u32(p.out + urow(i, j, k), lvec, f32x4_t{v[0], v[1], v[2], v[3]});
                }
                if (NRES > 0 && DEPTH == 1 && tl + 1 < TM * TN) issue(tl + 1, 0);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (VEC_FITS && full && (p.Cout & 3) == 0) {
            if (rb) epilogue_vec(I2{}); else if (ra) epilogue_vec(I1{}); else epilogue_vec(I0{});
        } else if (rb) { if (full) epilogue(I2{}, std::true_type{}); else epilogue(I2{}, std::false_type{}); }
        else if (ra) { if (full) epilogue(I1{}, std::true_type{}); else epilogue(I1{}, std::false_type{}); }
        else { if (full) epilogue(I0{}, std::true_type{}); else epilogue(I0{}, std::false_type{}); }
    }
}

// =================================================================================================================
// conv3x3_halo: 8 x 16 output pixels of one image x BN channels per block, halo patch per 32-channel block
// =================================================================================================================
// UP2: nearest-x2 upsample + conv as FOUR PHASE FILTERS of 2x2 taps on the low-resolution input (oracle/femasr_oracle.c
// orc_conv_up2_phases): a block takes an 8 x 16 LOW-resolution tile and one phase (a, b); output pixel (2y+a, 2x+b) reads
// patch rows y+a, y+a+1 and columns x+b, x+b+1 of the ordinary halo patch through the pre-summed weights p.w_up2[phase]
// (2.25x fewer MFMAs than sweeping the 9 taps over the upsampled image).
template <int BN, int WM, int WN, int PRO, bool UP2>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 4 : 2) void conv3x3_halo_kernel(const ConvParams p)
{
    constexpr int BM = 128, TW = 16;
    constexpr int PH = 10, PW = 18, PP = PH * PW;
    constexpr int NTAP = UP2 ? 4 : 9, SC = UP2 ? 2 : 1;
    static_assert(!(UP2 && PRO != FEMASR_PRO_NONE), "the phase kernel has no prologue");
    constexpr int NT = WM * WN * 64;
    constexpr int PUNITS = (PP * 8 + NT - 1) / NT;
    constexpr int PROWS = NT / 8;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1, "tile config");
    static_assert(PRO != FEMASR_PRO_LN, "no LayerNorm prologue on 3x3 convs");
    static_assert(2 + PUNITS <= BK / 2, "patch slices must fit the 16 k-pair steps of one tap");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PSZ = (((PP + 1) * ALD + 3) / 4) * 4;      // pixel PP is a write-only dummy slot (tail units)
    float *Ps = smem;               // [2][PP][ALD]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB;
    int tile = L / p.NB;
    const int pha = UP2 ? (tile >> 1) & 1 : 0, phb = UP2 ? tile & 1 : 0;       // phase innermost: the 4 phases of a tile share its patch in L2
    if (UP2) tile >>= 2;
    const int tx = tile % p.tilesX;
    tile /= p.tilesX;
    const int ty = tile % p.tilesY;
    const int n = tile / p.tilesY;
    const int oy0 = ty * 8, ox0 = tx * TW, n0 = nb * BN;       // tile origin in the conv's own pixel grid (low resolution when UP2)
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;

    // ---- this thread's patch units: (pixel = (t>>3) + PROWS i, channel quad kq)
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

    float4 rp[PUNITS], ga, gb;

    auto load_patch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < PUNITS; ++i) rp[i] = ld4(p.in + (size_t)poff[i] + (size_t)cc * BK);
        if (PRO == FEMASR_PRO_GN_SILU) {
            ga = ld4(p.pro_a + (size_t)n * p.Cin + cc * BK + 4 * kq);
            gb = ld4(p.pro_b + (size_t)n * p.Cin + cc * BK + 4 * kq);
        }
    };
    auto store_patch_unit = [&](int buf, int i) {
        float *Pb = Ps + buf * PSZ;
        const int pix = (t >> 3) + PROWS * i;
        if (PP % PROWS != 0 && i == PUNITS - 1 && pix >= PP) return;     // tail units of the last round
        float4 v = rp[i];
        if (PRO == FEMASR_PRO_GN_SILU) {     // two elements per instruction on the packed fp32 ALU (IEEE per component: same bits)
            const det_f32x2 s0 = det_silu2(__builtin_elementwise_fma(det_f32x2{v.x, v.y}, det_f32x2{ga.x, ga.y}, det_f32x2{gb.x, gb.y}));
            const det_f32x2 s1 = det_silu2(__builtin_elementwise_fma(det_f32x2{v.z, v.w}, det_f32x2{ga.z, ga.w}, det_f32x2{gb.z, gb.w}));
            v = make_float4(s0[0], s0[1], s1[0], s1[1]);
        }
        if (!(pmask & (1u << i))) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding AFTER the activation
        float *dst = Pb + pix * ALD + 4 * kq;
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const size_t wstride = (size_t)p.NT32 << 10;
    const float *wbase = UP2 ? p.w_up2 + (size_t)(pha * 2 + phb) * ((size_t)(p.Cin / BK) * NTAP) * wstride : p.w;
    const float *wl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        wl[j] = wbase + ((((size_t)wtile(n0, wn * TN + j, p.NT32)) * 64 + lane) << 4);

    const int ncc = p.Cin / BK;
    load_patch(0);
    float4 bc[TN], bn[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bc[j] = ld4(wl[j]);
#pragma unroll
    for (int i = 0; i < PUNITS; ++i) store_patch_unit(0, i);
    __syncthreads();

    // lane's output pixels: m_i = (wm*TM + i)*32 + (lane&31) -> (py, px) = (m >> 4, m & 15)
    int py[TM], px;
    {
        const int m = wm * TM * 32 + (lane & 31);
        px = m & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i) py[i] = (m >> 4) + 2 * i;
    }

    // LDS index of this lane's A-fragment pixel for (tap, row tile i): ONE per-lane base (which carries the phase offset)
    // plus a compile-time constant that folds into the ds_read offset.
    const int abase = ((py[0] + pha) * PW + px + phb) * ALD;
    auto patch_idx = [&](int tap, int (&idx)[TM]) {
        const int ky = UP2 ? tap >> 1 : tap / 3, kx = UP2 ? tap & 1 : tap - ky * 3;
#pragma unroll
        for (int i = 0; i < TM; ++i) idx[i] = abase + ((2 * i + ky) * PW + kx) * ALD;
    };

    // Main loop: unconditional straight-line code (9 taps x 16 k-pair steps unrolled; the last channel block re-stages
    // itself into the idle LDS buffer and re-reads the last weight chunk), so every s_waitcnt the compiler emits is exact.
    const int nq = ncc * NTAP;
    for (int cc = 0; cc < ncc; ++cc) {
        const float *Pb = Ps + (cc & 1) * PSZ + (lane >> 5);
        const int ccn = cc + 1 < ncc ? cc + 1 : cc;
        const bool more = cc + 1 < ncc;          // the LOADS stay unconditional (exact wait counters); normalise / activate / store only if needed
        const int nbuf = (cc + 1) & 1;
        float af[2][TM];
        int aidx[TM];
        patch_idx(0, aidx);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Pb[aidx[i]];
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int q = cc * NTAP + tap;
            const size_t qn = (size_t)(q + 1 < nq ? q + 1 : nq - 1);
            int nidx[TM];
            patch_idx(tap < NTAP - 1 ? tap + 1 : NTAP - 1, nidx);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1, g = kk >> 2, e = kk & 3;
                if (e == 0) {
                    if (g < 3) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) bn[j] = ld4(wl[j] + (size_t)q * wstride + 4 * (g + 1));
                    } else {
#pragma unroll
                        for (int j = 0; j < TN; ++j) bn[j] = ld4(wl[j] + qn * wstride);
                    }
                    // next channel block's patch: issued right AFTER a weight prefetch (loads complete in order; see
                    // conv_igemm_kernel), normalised / activated / stored one unit per k-pair step of tap 1
                    if (tap == 0 && kk == 0) load_patch(ccn);
                }
                if (kk + 1 < BK / 2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[nxt][i] = Pb[aidx[i] + 2 * (kk + 1)];
                } else if (tap < NTAP - 1) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[nxt][i] = Pb[nidx[i]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], f4get(bc[j], e), acc[i][j], 0, 0, 0);
                if (e == 3) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) bc[j] = bn[j];
                }
                if (tap == 1 && kk >= 2 && kk < 2 + PUNITS && more) store_patch_unit(nbuf, kk - 2);      // (uniform) nothing to stage after the last block
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) aidx[i] = nidx[i];
        }
        __syncthreads();     // patch buffers swap: ONE barrier per 32-channel block (NTAP taps x 16 MFMA steps)
    }

    // out = (acc + bias) + res1 + res2, in that order (bit-exact contract).  Address = uniform part (SGPRs) + one per-lane
    // offset: element r of row tile i sits at pixel row 2*(wm*TM+i) + (r>>3), pixel column (r&3) + 8*((r>>2)&1) +
    // 4*(lane>>5) of the 8x16 tile.  Full tiles: residual loads are branch-free batches of one 32x32 tile issued one tile
    // ahead of the stores (see conv_igemm_kernel's epilogue for why).
    // (UP2: tile pixel (y, x) is stored at (2y + a, 2x + b) of the (Ho, Wo) = (2H, 2W) output: every pixel step doubles)
    const int Hl = UP2 ? p.H : p.Ho, Wl = UP2 ? p.W : p.Wo;         // extent of the grid the tile indexes
    const bool full = (oy0 + 8 <= Hl) && (ox0 + TW <= Wl) && (n0 + BN <= p.Cout);
    const float *ra = p.res1 ? p.res1 : p.res2, *rb = (p.res1 && p.res2) ? p.res2 : nullptr;
    const size_t obase = (((size_t)n * p.Ho + SC * oy0 + pha) * p.Wo + SC * ox0 + phb) * p.Cout + n0;             // uniform
    const unsigned loff = (unsigned)(SC * 4 * (lane >> 5)) * (unsigned)p.Cout + (unsigned)(lane & 31);
    const int wmu = __builtin_amdgcn_readfirstlane(wm), wnu = __builtin_amdgcn_readfirstlane(wn);    // provably uniform copies
    auto uoff = [&](int i, int j, int r) -> size_t {            // uniform
        return obase + (size_t)(SC * ((2 * (wmu * TM + i) + (r >> 3)) * p.Wo + (r & 3) + 8 * ((r >> 2) & 1))) * p.Cout + (wnu * TN + j) * 32;
    };
    // Fused GroupNorm(32) partial moments of the OUTPUT (consumed by the next conv's GN prologue): fp64 sums of the stored
    // values in the fixed order of oracle/femasr_oracle.c orc_gn_coeffs - per lane over its 16 accumulator registers
    // (level 0), the two lane halves (1), the channels of the group (2), the tile's four 32-pixel blocks (3).
    const bool gnp = p.gn_part != nullptr;               // (UP2: the partial of this half-resolution tile and phase, index tile*4 + phase)
    double gs[TM][TN], gss[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { gs[i][j] = 0.0; gss[i][j] = 0.0; }
    // edge tiles: the same code with clamped addresses (uniform part -> this tile's first element, lane part -> 0) for the
    // loads and masked stores
    auto epilogue = [&](auto nres_c, auto full_c) {
        constexpr int NRES = decltype(nres_c)::value;
        constexpr bool FULL = decltype(full_c)::value;
        constexpr int DEPTH = NRES == 1 ? 2 : 1;         // one residual: loads run a tile ahead; two: per-tile batches
        float rbuf[DEPTH][NRES > 0 ? NRES : 1][16];
        auto ok_u = [&](int i, int j, int r) -> bool {
            return FULL || ((oy0 + 2 * (wmu * TM + i) + (r >> 3)) < Hl && (ox0 + (r & 3) + 8 * ((r >> 2) & 1)) < Wl &&
                            (n0 + (wnu * TN + j) * 32) < p.Cout);
        };
        auto ok_l = [&](int i, int j, int r) -> bool {
            return FULL || ((oy0 + 2 * (wmu * TM + i) + (r >> 3)) < Hl &&
                            (ox0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * (lane >> 5)) < Wl &&
                            (n0 + (wnu * TN + j) * 32 + (lane & 31)) < p.Cout);
        };
        auto issue = [&](int tl, int slot) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = tl / TN, j = tl % TN;
                const size_t ou = ok_u(i, j, r) ? uoff(i, j, r) : obase;
                const unsigned lo = ok_l(i, j, r) ? 4u * loff : 0u;
                if (NRES >= 1) rbuf[slot][0][r] = ldg_u32(ra + ou, lo);
                if (NRES >= 2) rbuf[slot][NRES >= 2 ? 1 : 0][r] = ldg_u32(rb + ou, lo);
            }
        };
        if (NRES > 0) issue(0, 0);
#pragma unroll
        for (int tl = 0; tl < TM * TN; ++tl) {
            const int i = tl / TN, j = tl % TN;
            if (DEPTH == 2 && tl + 1 < TM * TN) issue(tl + 1, (tl + 1) & 1);
            const int col = n0 + (wnu * TN + j) * 32 + (lane & 31);
            const float bv = (FULL || col < p.Cout) ? p.bias[FULL ? col : (col < p.Cout ? col : 0)] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv;
                if (NRES >= 1) v = v + rbuf[tl % DEPTH][0][r];
                if (NRES >= 2) v = v + rbuf[tl % DEPTH][NRES >= 2 ? 1 : 0][r];
                if (ok_l(i, j, r)) {
                    stg_u32(p.out + uoff(i, j, r), 4u * loff, v);
                    if (gnp) {
                        const double d = (double)v;
                        gs[i][j] = gs[i][j] + d;
                        gss[i][j] = __builtin_fma(d, d, gss[i][j]);
                    }
                }
            }
            if (NRES > 0 && DEPTH == 1 && tl + 1 < TM * TN) issue(tl + 1, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (rb) { if (full) epilogue(I2{}, std::true_type{}); else epilogue(I2{}, std::false_type{}); }
    else if (ra) { if (full) epilogue(I1{}, std::true_type{}); else epilogue(I1{}, std::false_type{}); }
    else { if (full) epilogue(I0{}, std::true_type{}); else epilogue(I0{}, std::false_type{}); }

    if (gnp) {      // uniform
        const int cg = p.Cout >> 5;                          // channels per group (power of two <= 32, checked by the launcher)
        const int gpb = BN / cg;                             // groups per block (<= 64 here: BN = 32 only serves Cout <= 32)
        double *red = reinterpret_cast<double *>(smem);      // [4 (q)][gpb][2]; the patch buffers are dead (last loop barrier)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                double a = gs[i][j] + __shfl_xor(gs[i][j], 32, 64);          // level 1: lane halves (commutative: same bits in both)
                double b = gss[i][j] + __shfl_xor(gss[i][j], 32, 64);
                for (int d = 1; d < cg; d <<= 1) {                            // level 2: xor-butterfly over the group's channels
                    a = a + __shfl_xor(a, d, 64);
                    b = b + __shfl_xor(b, d, 64);
                }
                const int cl = (wnu * TN + j) * 32 + (lane & 31);             // channel inside the block
                if (lane < 32 && (cl & (cg - 1)) == 0) {
                    double *dst = red + ((size_t)(wmu * TM + i) * gpb + cl / cg) * 2;
                    dst[0] = a;
                    dst[1] = b;
                }
            }
        __syncthreads();
        if (t < gpb) {
            const int g = n0 / cg + t;
            if (g < 32) {
                double S = red[(0 * gpb + t) * 2], SS = red[(0 * gpb + t) * 2 + 1];
#pragma unroll
                for (int q = 1; q < 4; ++q) {                                // level 3: ((q0 + q1) + q2) + q3
                    S = S + red[((size_t)q * gpb + t) * 2];
                    SS = SS + red[((size_t)q * gpb + t) * 2 + 1];
                }
                const size_t tidx = (size_t)n * p.tilesY * p.tilesX + (size_t)ty * p.tilesX + tx;
                double *dst = p.gn_part + ((UP2 ? tidx * 4 + (pha * 2 + phb) : tidx) * 32 + g) * 2;
                dst[0] = S;
                dst[1] = SS;
            }
        }
    }
}

// =================================================================================================================
// conv3x3_cout3: 3x3 stride-1 pad-1 conv with 3 output channels (out_conv, femasr_arch.py:273) on the VALU.
// An MFMA tile would compute 32 output columns to keep 3 (10.7x the algorithmic work: 1.7 ms of the B = 16 step); here one
// thread owns one pixel and its 3 fmaf chains (same k order as everywhere: 32-channel blocks outermost, then the 9 taps),
// activations come from an LDS halo patch as 16-byte reads (pitch 36 floats: conflict-free), the weights sit in SGPRs
// (compact [k][4] array read with scalar loads; `v_fmac_f32 v, s, v`).  8 x 32 pixels per block.
// =================================================================================================================
constexpr int C3_TH = 8, C3_TW = 32, C3_PW = C3_TW + 2, C3_PP = (C3_TH + 2) * C3_PW, C3_PITCH = 36;

__global__ __launch_bounds__(256) void conv3x3_cout3_kernel(const ConvParams p, const float *__restrict__ wc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [C3_PP][C3_PITCH]
    const int t = threadIdx.x;
    int tile = blockIdx.x;
    const int tx = tile % p.tilesX;
    tile /= p.tilesX;
    const int ty = tile % p.tilesY;
    const int n = tile / p.tilesY;
    const int oy0 = ty * C3_TH, ox0 = tx * C3_TW;
    const int py = t >> 5, px = t & 31;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    const int ncb = p.Cin >> 5;
    // this thread's patch units (pixel, channel quad): fixed per tile; the next channel block's units are fetched into
    // registers BEFORE the current block's FMAs and written to LDS after them (one buffer, loads in flight under the math)
    constexpr int NU = (C3_PP * 8 + 255) / 256;
    unsigned uoff[NU];
    unsigned umask = 0;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = t + 256 * i, pix = u >> 3, kq = u & 7;
        const int sy = oy0 - 1 + pix / C3_PW, sx = ox0 - 1 + pix % C3_PW;
        const bool ok = u < C3_PP * 8 && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
        uoff[i] = ok ? (unsigned)((((size_t)n * p.H + sy) * p.W + sx) * p.Cin + 4 * kq) : 0u;
        umask |= (ok ? 1u : 0u) << i;
    }
    float4 ru[NU];
    auto fetch = [&](int cb) {
#pragma unroll
        for (int i = 0; i < NU; ++i) ru[i] = ld4(p.in + (size_t)uoff[i] + cb * 32);
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = t + 256 * i;
            if (u < C3_PP * 8)
                *reinterpret_cast<float4 *>(smem + (u >> 3) * C3_PITCH + 4 * (u & 7)) = (umask >> i) & 1u ? ru[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int cb = 0; cb < ncb; ++cb) {
        if (cb) __syncthreads();          // every wave is done with the previous block's patch
        stage();
        __syncthreads();
        fetch(cb + 1 < ncb ? cb + 1 : cb);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float *src = smem + ((py + tap / 3) * C3_PW + px + tap % 3) * C3_PITCH;
            const float *w = wc + ((size_t)(cb * 9 + tap) * 32) * 4;      // uniform: scalar loads
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
                const float4 x4 = *reinterpret_cast<const float4 *>(src + 4 * c4);
                const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float *wk = w + (4 * c4 + e) * 4;
                    acc0 = __builtin_fmaf(xs[e], wk[0], acc0);
                    acc1 = __builtin_fmaf(xs[e], wk[1], acc1);
                    acc2 = __builtin_fmaf(xs[e], wk[2], acc2);
                }
            }
        }
    }
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy < p.Ho && ox < p.Wo) {
        const size_t o = (((size_t)n * p.Ho + oy) * p.Wo + ox) * 3;
        float v[3] = {acc0 + p.bias[0], acc1 + p.bias[1], acc2 + p.bias[2]};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (p.res1) v[c] = v[c] + p.res1[o + c];
            if (p.res2) v[c] = v[c] + p.res2[o + c];
            p.out[o + c] = v[c];
        }
    }
}

template <bool UP2>
constexpr size_t halo_lds_bytes()
{
    return (size_t)(2 * ((((180 + 1) * ALD + 3) / 4) * 4)) * sizeof(float);
}

template <int BM>
constexpr size_t conv_lds_bytes() { return (size_t)(2 * BM * ALD) * sizeof(float); }

struct Variant {
    const char *name;
    int bm, bn;
    void (*kern)(const ConvParams);
    size_t lds;
    unsigned long long attr_devs;       // bit d: MaxDynamicSharedMemorySize set on device d (the attribute is per device)
    int threads;
};

#define FEMASR_VARIANT(BM, BN, WM, WN, PRO, VEC)                                                            \
    { "conv_igemm<" #BM "x" #BN "," #PRO ",cinvec=" #VEC ",waves=" #WM "x" #WN ">",                         \
      BM, BN, conv_igemm_kernel<BM, BN, WM, WN, PRO, VEC>, conv_lds_bytes<BM>(), 0ull, WM * WN * 64 }
#define FEMASR_HALO(BN, WM, WN, PRO, UP2)                                                                  \
    { "conv3x3_halo<8x16x" #BN "," #PRO ",up2=" #UP2 ",waves=" #WM "x" #WN ">", 128, BN,                   \
      conv3x3_halo_kernel<BN, WM, WN, PRO, UP2>, halo_lds_bytes<UP2>(), 0ull, WM * WN * 64 }

Variant g_variants[] = {
    FEMASR_VARIANT(128, 128, 4, 2, FEMASR_PRO_NONE, true),       // 0  general convs, Cin % 32 == 0 (stride-2 convs): cls*2 + pro
    FEMASR_VARIANT(128, 128, 4, 2, FEMASR_PRO_GN_SILU, true),    // 1
    FEMASR_VARIANT(128, 64, 4, 2, FEMASR_PRO_NONE, true),        // 2
    FEMASR_VARIANT(128, 64, 4, 2, FEMASR_PRO_GN_SILU, true),     // 3
    FEMASR_VARIANT(128, 32, 4, 1, FEMASR_PRO_NONE, true),        // 4
    FEMASR_VARIANT(128, 32, 4, 1, FEMASR_PRO_GN_SILU, true),     // 5
    FEMASR_VARIANT(128, 128, 4, 2, FEMASR_PRO_NONE, false),      // 6  generic Cin (in_conv): 6 + cls
    FEMASR_VARIANT(128, 64, 4, 2, FEMASR_PRO_NONE, false),       // 7
    FEMASR_VARIANT(128, 32, 4, 1, FEMASR_PRO_NONE, false),       // 8
    FEMASR_HALO(128, 2, 2, FEMASR_PRO_NONE, false),              // 9  3x3 s1 halo kernels, 4 waves of 64 px x 64 ch: 9 + cls*3 + (up2 ? 2 : pro)
    FEMASR_HALO(128, 2, 2, FEMASR_PRO_GN_SILU, false),           // 10
    FEMASR_HALO(128, 2, 2, FEMASR_PRO_NONE, true),               // 11 nearest-x2 + conv as 4 phase filters of 2x2 taps
    FEMASR_HALO(64, 2, 2, FEMASR_PRO_NONE, false),               // 12 (64 px x 32 ch per wave)
    FEMASR_HALO(64, 2, 2, FEMASR_PRO_GN_SILU, false),            // 13
    FEMASR_HALO(64, 2, 2, FEMASR_PRO_NONE, true),                // 14
    FEMASR_HALO(32, 4, 1, FEMASR_PRO_NONE, false),               // 15 (out_conv, Cout = 3): 4 waves of 32 px x 32 ch
    FEMASR_HALO(32, 4, 1, FEMASR_PRO_GN_SILU, false),            // 16
    FEMASR_HALO(32, 4, 1, FEMASR_PRO_NONE, true),                // 17
    { "conv3x3_cout3<8x32,valu>", 256, 4, nullptr, (size_t)C3_PP * C3_PITCH * sizeof(float), 0ull, 256 },      // 18 out_conv (own launcher branch)
};
constexpr int kNumVariants = sizeof(g_variants) / sizeof(g_variants[0]);
constexpr int kFirstHalo = 9;

std::atomic<int> g_small_blocks{-1};         // -1: not set (FEMASR_CONV_SMALL_BLOCKS or the default); test hook, process-global
// default threshold: 1.5 x the device's CUs (384 on the 256-CU MI355X in SPX mode; a partition with fewer CUs gets its own value)
int small_blocks_default()
{
    static const int v = [] {
        const char *e = getenv("FEMASR_CONV_SMALL_BLOCKS");
        if (e && atoi(e) >= 0) return atoi(e);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            return cus + cus / 2;
        return 384;
    }();
    return v;
}
int small_launch_blocks()
{
    const int v = g_small_blocks.load(std::memory_order_relaxed);
    return v >= 0 ? v : small_blocks_default();
}

int pick_variant(const femasr_conv_args *a, int Ho, int Wo)
{
    const bool vec = (a->Cin % BK) == 0;
    int cls = a->Cout > 64 ? 0 : (a->Cout > 32 ? 1 : 2);           // BN = 128 / 64 / 32
    const bool halo = femasr_conv_halo_eligible(a);
    // Small launches (batch 1-2 of the 72x72 / 144x144 layers): with 128-column blocks a 256-channel layer of one tile is 90
    // blocks on 256 CUs, each a serial chain over 9 taps x Cin.  64-column blocks double the block count and halve the chain
    // (same fragment-major weights, same per-wave pixel tile: results and GroupNorm partial moments are unchanged).
    if (cls == 0 && (halo || vec)) {
        const long long mb = halo ? (long long)a->B * (((a->up2 ? a->W : Wo) + 15) / 16) * (((a->up2 ? a->H : Ho) + 7) / 8) * (a->up2 ? 4 : 1)
                                  : ((long long)a->B * Ho * Wo + 127) / 128;
        if (mb * ((a->Cout + 127) / 128) < small_launch_blocks()) cls = 1;
    }
    // Cout > 32: 4-wave halo blocks (64 px x 64 / 32 ch per wave: half the LDS / weight-fragment reads per MFMA, 4 waves per
    // barrier instead of 8; measured +4..9 % over the 8-wave tiling, fused-x2 variant 141 TFLOP/s = 90 % of peak)
    if (halo) return kFirstHalo + cls * 3 + (a->up2 ? 2 : a->prologue);
    if (!vec) return 6 + cls;
    return cls * 2 + a->prologue;
}

}  // namespace

extern "C" int femasr_conv_small_launch_blocks(int blocks)
{
    const int prev = small_launch_blocks();
    g_small_blocks.store(blocks >= 0 ? blocks : -1, std::memory_order_relaxed);
    return prev;
}

// 3x3 stride-1 pad-1 convs with Cin % 32 == 0 run on the halo kernels (shared with model.hip's planner)
bool femasr_conv_halo_eligible(const femasr_conv_args *a)
{
    return a->ksz == 3 && a->stride == 1 && a->pad == 1 && (a->Cin % BK) == 0 && a->prologue != FEMASR_PRO_LN &&
           a->act == FEMASR_ACT_NONE && !(a->up2 && a->prologue != FEMASR_PRO_NONE) &&
           (size_t)a->B * a->H * a->W * a->Cin < ((size_t)1 << 31);
}

int femasr_conv_variant_count() { return kNumVariants + femasr_gemm_variant_count(); }
const char *femasr_conv_variant_name(int v)
{
    if (v >= kNumVariants) return femasr_gemm_variant_name(v - kNumVariants);
    return v >= 0 ? g_variants[v].name : "?";
}

int femasr_conv2d_launch(hipStream_t s, const femasr_conv_args *a, const conv_vq_epilogue *vq,
                         int *variant_out, double *flops_out)
{
    FEMASR_REQUIRE(a && a->in && a->w && a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0,
                   "conv2d: null pointer or empty shape");
    FEMASR_REQUIRE(vq || (a->bias && a->out), "conv2d: bias/out must be set");
    FEMASR_REQUIRE(a->ksz >= 1 && a->ksz <= 7 && (a->stride == 1 || a->stride == 2) && a->pad >= 0,
                   "conv2d: unsupported ksz=%d stride=%d pad=%d", a->ksz, a->stride, a->pad);
    FEMASR_REQUIRE(a->prologue == FEMASR_PRO_NONE || a->prologue == FEMASR_PRO_GN_SILU,
                   "conv2d: bad prologue %d (LayerNorm is a separate pass: femasr_layernorm)", a->prologue);
    const int Hv = a->up2 ? 2 * a->H : a->H, Wv = a->up2 ? 2 * a->W : a->W;
    const int Ho = (Hv + 2 * a->pad - a->ksz) / a->stride + 1, Wo = (Wv + 2 * a->pad - a->ksz) / a->stride + 1;
    FEMASR_REQUIRE(Ho == a->Ho && Wo == a->Wo, "conv2d: Ho/Wo mismatch (%d,%d) vs expected (%d,%d)", a->Ho, a->Wo, Ho, Wo);
    // 1x1 convs / nn.Linear / the VQ distance matrix: the LDS-DMA GEMM (kernels_gemm.hip); its weights are packed by
    // femasr_repack_oihw in the k-permuted layout that kernel consumes
    if (femasr_gemm_eligible(a)) {
        int gv = 0;
        const int rc = femasr_gemm_launch(s, a, vq, &gv, flops_out);
        if (variant_out) *variant_out = kNumVariants + gv;
        return rc;
    }
    FEMASR_REQUIRE(!vq, "conv2d: the VQ epilogue needs a 1x1 layer with Cin %% 32 == 0");
    const bool vec = (a->Cin % BK) == 0;
    FEMASR_REQUIRE(vec || a->prologue == FEMASR_PRO_NONE, "conv2d: prologue needs Cin %% 32 == 0 (Cin=%d)", a->Cin);
    if (a->prologue == FEMASR_PRO_GN_SILU) FEMASR_REQUIRE(a->pro_a && a->pro_b, "conv2d: GN prologue needs a,b");
    const long long M = (long long)a->B * Ho * Wo;
    FEMASR_REQUIRE(M < (1ll << 31) - 256, "conv2d: too many output pixels");

    ConvParams p{};
    p.in = a->in; p.w = a->w; p.bias = a->bias; p.pro_a = a->pro_a; p.pro_b = a->pro_b; p.pro_c = a->pro_c;
    p.res1 = a->res1; p.res2 = a->res2; p.out = a->out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.ksz = a->ksz; p.stride = a->stride;
    p.pad = a->pad; p.up2 = a->up2; p.act = a->act; p.Ho = Ho; p.Wo = Wo;
    p.M = (int)M; p.K = a->ksz * a->ksz * a->Cin; p.nchunks = (p.K + BK - 1) / BK; p.taps = a->ksz * a->ksz;
    p.NT32 = (a->Cout + 31) / 32;
    p.gn_part = a->gn_part;
    p.kperm = (a->ksz == 1 && vec) ? 1 : 0;       // weights packed by femasr_repack_oihw in the GEMM layout
    if (femasr_conv_halo_eligible(a) && a->Cout == 3 && !a->up2 && a->prologue == FEMASR_PRO_NONE && !a->gn_part) {
        // out_conv: direct VALU kernel on the compact [k][4] weights stored behind the fragment-major matrix
        const size_t tail = femasr_compact_weight_floats(a->Cout, a->Cin, 3, 3);
        const float *wc = a->w + (size_t)p.nchunks * p.NT32 * 1024;
        p.tilesX = (Wo + C3_TW - 1) / C3_TW;
        p.tilesY = (Ho + C3_TH - 1) / C3_TH;
        (void)tail;
        hipLaunchKernelGGL(conv3x3_cout3_kernel, dim3((unsigned)(a->B * p.tilesX * p.tilesY)), dim3(256), (size_t)C3_PP * C3_PITCH * sizeof(float), s,
                           p, wc);
        FEMASR_CHECK_HIP(hipGetLastError());
        if (variant_out) *variant_out = kNumVariants - 1;
        if (flops_out) *flops_out = 2.0 * (double)M * (double)a->Cout * (double)p.K;
        return FEMASR_OK;
    }
    const int vi = pick_variant(a, Ho, Wo);
    Variant &v = g_variants[vi];
    p.MB = (p.M + v.bm - 1) / v.bm;
    p.NB = (p.Cout + v.bn - 1) / v.bn;
    const bool phases = vi >= kFirstHalo && a->up2;
    if (vi >= kFirstHalo) {   // halo kernels: 2-D tiles of 8 x 16 pixels per image (of the low-resolution grid x 4 phases when up2)
        p.tilesX = ((phases ? a->W : Wo) + 15) / 16;
        p.tilesY = ((phases ? a->H : Ho) + 7) / 8;
        p.MB = a->B * p.tilesX * p.tilesY * (phases ? 4 : 1);
    }
    FEMASR_REQUIRE(!phases || a->w_up2, "conv2d: a 3x3 nearest-x2 conv with Cin %% 32 == 0 needs w_up2 (femasr_repack_oihw_up2)");
    p.w_up2 = a->w_up2;
    FEMASR_REQUIRE(!a->gn_part || (vi >= kFirstHalo && femasr_gn_fusable(a->Cout)),
                   "conv2d: gn_part (fused GroupNorm partial moments) needs a 3x3 stride-1 halo conv and 32 | Cout, Cout/32 a power of two <= 32");
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((__atomic_load_n(&v.attr_devs, __ATOMIC_ACQUIRE) >> dev) & 1ull)) {      // (idempotent: a race only repeats the call)
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
        if (dev >= 0 && dev < 64) __atomic_fetch_or(&v.attr_devs, 1ull << dev, __ATOMIC_RELEASE);
    }
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(p.MB * p.NB)), dim3((unsigned)v.threads), v.lds, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (variant_out) *variant_out = vi;
    // ALGORITHMIC flops (the definition: 9 taps per output pixel); the phase form of an x2 conv issues 4/9 of them
    if (flops_out) *flops_out = 2.0 * (double)M * (double)a->Cout * (double)p.K;
    return FEMASR_OK;
}
