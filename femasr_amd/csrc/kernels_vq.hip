// kernels_vq.hip — the codebook lookup (VectorQuantizer.forward, femasr_arch.py:35-38,50-100) as a two-pass EXACT search.
//
// The specification (DESIGN.md "Arithmetic specification", oracle/femasr_oracle.c orc_vq) is
//     d[m][j] = (zz[m] + ee[j]) - 2 * dot(m, j),   dot = ONE fp32 fmaf chain over c in ORC_KPERM order,
//     idx[m]  = first minimum of d[m][:]
// and the single-pass path evaluates all M x n_e chains on the fp32 MFMA (kernels_gemm.hip, vq=true).  Here:
//
//   pass 1  vq_candidates_kernel   bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16x the fp32 rate) computes approximate dots and,
//                                  from a RIGOROUS bound on |approximate - specified| distance, keeps for every row the
//                                  codes whose lower bound does not exceed the row's smallest upper bound.  The exact
//                                  first-min is always among them (a code outside the set is strictly farther than
//                                  some code inside it).
//   pass 2  vq_exact_kernel        evaluates the specified fp32 chain for the surviving (row, code) pairs on the VALU,
//                                  takes the first-min with the same tie rule, writes idx and z_q = z + (e - z).
//
// Error bound used by pass 1 (K = e_dim <= 512 products per dot):
//   bf16 RNE rounding of both operands       |z~e~ - ze| <= (2^-8 + 2^-18) |z||e|      per product
//   fp32 accumulation inside the MFMA        <= K * 2^-23 * sum|z~e~|                  (any order, any rounding mode)
//   the specified chain vs the real dot      <= K * 2^-24 * sum|ze|
//   sum|ze| <= |z| |e| (Cauchy-Schwarz)  =>  |dot~ - dot_spec| <= 0.004004 |z| |e|;  VQ_EPS = 0.0042 with
//   |e| <= max_j sqrt(ee_j) * 1.001 (one bound per row: the largest code norm) and |z| <= |z~| * 1.004 (the norm of the
//   bf16 image in LDS; |z~| >= |z| (1 - 2^-9)).  d differs from (ee - 2 dot~) + zz by at most 2*that plus three fp32
//   roundings of magnitude <= 2^-24 (zz + ee + 2|dot|), covered by the per-row slack gz = 2^-20 (zz + ee_max + 2|z||e|max).
//   A code survives iff  s_j <= min_j s_j + 2 E_row,  s = ee - 2 dot~,  E_row = 2 VQ_EPS |z| |e|max + gz.
//
// Rows whose candidate list overflows (more than VQ_CMAX = 32 entries) are marked VQ_ALL and pass 2 evaluates every code
// for them: slow, still exact.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int VQ_R = 128;                  // rows per block of pass 1 (4 MFMA column tiles)
constexpr int VQ_CMAX = 32;                // candidate slots per row (LDS list of pass 1 = the list handed to pass 2)
constexpr int VQ_ALL = 0xFFFF;             // count marker: evaluate every code
constexpr float VQ_EPS = 0.0042f;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));      // low half = a, high half = b, RNE
    return r;
}
__device__ __forceinline__ uint4 pack8(const float4 &a, const float4 &b)
{
    return uint4{pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w)};
}

// codebook (n_e, D) fp32 -> MFMA A-operand image, bf16:  out[ct][s][lane] = 8 consecutive k of code 32 ct + lane%32,
// k = 16 s + 8 (lane/32) ..+7;  en[j] = upper bound of |e_j|.
__global__ void vq_pack_codebook_kernel(const float *__restrict__ cb, const float *__restrict__ ee, int n_e, int D,
                                        uint4 *__restrict__ out, float *__restrict__ en)
{
    const int S = D >> 4;
    const size_t total = (size_t)(n_e >> 5) * S * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int s = (int)((i >> 6) % S), ct = (int)((i >> 6) / S);
        const float *src = cb + (size_t)(ct * 32 + (lane & 31)) * D + 16 * s + 8 * (lane >> 5);
        out[i] = pack8(ld4(src), ld4(src + 4));
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_e; j += gridDim.x * blockDim.x) en[j] = sqrtf(ee[j]) * 1.001f;
}

// en[n_e] = max ee, en[n_e + 1] = max en (finalize time, one thread)
__global__ void vq_codebook_max_kernel(const float *__restrict__ ee, int n_e, float *__restrict__ en)
{
    if (blockIdx.x || threadIdx.x) return;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < n_e; ++j) {
        a = fmaxf(a, ee[j]);
        b = fmaxf(b, en[j]);
    }
    en[n_e] = a;
    en[n_e + 1] = b;
}

struct VqCandParams {
    const float *z;
    long long M;
    int D, n_e;
    const uint4 *cbp;
    const float *ee, *en;           // en[n_e], en[n_e + 1] = max ee, max en
    unsigned short *cand, *cnt;     // !FUSED: the candidate lists, handed to vq_exact_kernel through global memory
    const float *cb;                // FUSED: the fp32 codebook and the outputs of the lookup
    long long *idx;
    float *zq;
};

__device__ __forceinline__ uint4 ld_code(const uint4 *cbp, int S, int ct, int s, int lane) { return cbp[((size_t)ct * S + s) * 64 + lane]; }

// pass 2: lane = (row, candidate slot); the specified fp32 chain (ORC_KPERM order inside every 8 channels), first-min
// over the row's candidates with ties to the smaller code, idx and z_q = z + (e[idx] - z).  |z|^2 (one chain, c ascending,
// as femasr_row_sqsum) is accumulated alongside the first chain of the row.  The chains are latency-bound (each lane
// streams its own two 2-KB rows): 32-float bursts = one 128-B line per operand, double-buffered.
struct VqBurst {
    float4 z[8], e[8];
};
__device__ __forceinline__ void vq_load_burst(VqBurst &b, const float *zr, const float *er, int c)
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        b.z[u] = ld4(zr + c + 4 * u);
        b.e[u] = ld4(er + c + 4 * u);
    }
}
__device__ __forceinline__ void vq_chain_burst(const VqBurst &b, float &acc, float &zacc)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 z0 = b.z[2 * g], z1 = b.z[2 * g + 1], e0 = b.e[2 * g], e1 = b.e[2 * g + 1];
        acc = __builtin_fmaf(z0.x, e0.x, acc);
        acc = __builtin_fmaf(z1.x, e1.x, acc);
        acc = __builtin_fmaf(z0.y, e0.y, acc);
        acc = __builtin_fmaf(z1.y, e1.y, acc);
        acc = __builtin_fmaf(z0.z, e0.z, acc);
        acc = __builtin_fmaf(z1.z, e1.z, acc);
        acc = __builtin_fmaf(z0.w, e0.w, acc);
        acc = __builtin_fmaf(z1.w, e1.w, acc);
        zacc = __builtin_fmaf(z0.x, z0.x, zacc);
        zacc = __builtin_fmaf(z0.y, z0.y, zacc);
        zacc = __builtin_fmaf(z0.z, z0.z, zacc);
        zacc = __builtin_fmaf(z0.w, z0.w, zacc);
        zacc = __builtin_fmaf(z1.x, z1.x, zacc);
        zacc = __builtin_fmaf(z1.y, z1.y, zacc);
        zacc = __builtin_fmaf(z1.z, z1.z, zacc);
        zacc = __builtin_fmaf(z1.w, z1.w, zacc);
    }
}
// both chains of one (row, code) pair; D % 32 == 0.  (A double-buffered variant needs 128 burst registers and spilled
// at any occupancy worth having; single bursts with 4+ waves per SIMD measured faster.)
__device__ __forceinline__ void vq_chain(const float *zr, const float *er, int D, float &acc, float &zacc)
{
    for (int c = 0; c < D; c += 32) {
        VqBurst b;
        vq_load_burst(b, zr, er, c);
        vq_chain_burst(b, acc, zacc);
    }
}

// The same two chains with the row of z in LDS (fp32, 16-byte chunks xor-swizzled by the row: the 8 rows a wave reads together sit in
// different banks) and the code's row streamed from L2 one 128-byte burst ahead.  D % 64 == 0.  Operation order = vq_chain's.
// (Three bursts ahead - 128 registers of code rows per lane - measured SLOWER: 336 against 311 us for the whole lookup at M = 82 944; every
// lane walks its own 2-KB row, 16 bytes per request and line: the phase is bound by the texture addresser's line rate, not by latency.)
__device__ __forceinline__ void vq_load_e(float4 (&e)[8], const float *er, int c)
{
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = ld4(er + c + 4 * u);
}
__device__ __forceinline__ void vq_chain_lds(const float *zrow, int sw, const float *er, int D, float &acc, float &zacc)
{
    float4 ea[8], eb[8];
    vq_load_e(ea, er, 0);
    for (int c = 0; c < D; c += 64) {
        vq_load_e(eb, er, c + 32);
        {
            VqBurst b;
#pragma unroll
            for (int u = 0; u < 8; ++u) { b.z[u] = ld4(zrow + c + 4 * (u ^ sw)); b.e[u] = ea[u]; }
            vq_chain_burst(b, acc, zacc);
        }
        if (c + 64 < D) vq_load_e(ea, er, c + 64);
        {
            VqBurst b;
#pragma unroll
            for (int u = 0; u < 8; ++u) { b.z[u] = ld4(zrow + c + 32 + 4 * (u ^ sw)); b.e[u] = eb[u]; }
            vq_chain_burst(b, acc, zacc);
        }
    }
}

// pass 1.  Block = NW waves (2 per SIMD: one wave's bound arithmetic overlaps the other's MFMAs), 128 rows of z as a bf16
// image in LDS (the MFMA B operand); wave w takes code-tile pairs w, w+NW, ..: per 16-deep k step 2 A fragments from global
// (L2-resident packed codebook, 4-step register ring), 4 B fragments from LDS, 8 MFMAs.
// Bound per row (see the header): every code's |d - zz - s| <= Erow, s = ee - 2 dot~, Erow = 2 VQ_EPS |z| |e|max + gz;
// a code survives iff s <= min_j s_j + 2 Erow.
// FUSED (round 4, the default): the block also runs the EXACT phase for its 128 rows - the lists never leave LDS, idx and z_q are written
// here, ONE launch.  The bf16 image is dead after the search: its 128 KB take the fp32 rows of 64 rows at a time (re-read from L2 / MALL,
// where this block's first read put them), lane = (row, candidate slot) walks the specified chains with z from LDS and the code's row
// streamed from L2, and z_q = z + (e - z) is formed from the LDS copy.
template <int NW, bool FUSED>
__global__ __launch_bounds__(NW * 64, NW / 4) void vq_candidates_kernel(const VqCandParams p)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, c31 = lane & 31, h = lane >> 5;
    const int D = p.D, CH = D >> 3, S = D >> 4, SWM = (CH < 16 ? CH : 16) - 1, LCH = 31 - __builtin_clz(CH);
    uint4 *zt = reinterpret_cast<uint4 *>(smem_raw);                               // [128][CH] 16-B chunks, XOR-swizzled
    float *s_ee = reinterpret_cast<float *>(smem_raw + (size_t)VQ_R * D * 2);
    // (an explicit LDS pointer: through a generic `volatile float *` the address space is not inferred and every access becomes a flat one
    // with a 64-bit address)
    typedef __attribute__((address_space(3))) volatile float lds_vfloat;
    typedef __attribute__((address_space(3))) const volatile f32x4_t lds_vfloat4;
    lds_vfloat *s_min = (lds_vfloat *)(s_ee + p.n_e);                                          // [128][NW] running min of s, read without barriers
    float *s_e2 = s_ee + p.n_e + NW * VQ_R;                                        // [128] |z~|^2, then 2 Erow
    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_e2 + VQ_R);
    unsigned short *s_code = reinterpret_cast<unsigned short *>(s_cnt + VQ_R);     // [128][VQ_CMAX]

    const long long row0 = (long long)blockIdx.x * VQ_R;
    for (int i = t; i < p.n_e; i += NW * 64) s_ee[i] = p.ee[i];
    for (int i = t; i < NW * VQ_R; i += NW * 64) s_min[i] = INFINITY;
    if (t < VQ_R) { s_cnt[t] = 0u; s_e2[t] = 0.f; }
    __syncthreads();
    // 8 chunks (16 float4 loads) in flight per thread: the block is alone on its CU, latency is hidden by depth only
    for (int i0 = t; i0 < VQ_R * CH; i0 += NW * 64 * 8) {
        float4 va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * NW * 64, r = i >> LCH, c = i & (CH - 1);
            va[u] = vb[u] = float4{0.f, 0.f, 0.f, 0.f};
            if (i < VQ_R * CH && row0 + r < p.M) {
                const float *src = p.z + (size_t)(row0 + r) * D + c * 8;
                va[u] = ld4(src);
                vb[u] = ld4(src + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * NW * 64, r = i >> LCH, c = i & (CH - 1);
            if (i < VQ_R * CH) zt[r * CH + (c ^ (r & SWM))] = pack8(va[u], vb[u]);
        }
    }
    __syncthreads();
    {   // |z~|^2 per row from the bf16 image (bound only: any order).  4 threads per row, interleaved chunks.
        const int r = t >> 2, part = t & 3;
        float q = 0.f;
        if (r < VQ_R) {
            for (int c = part; c < CH; c += 4) {
                const uint4 v = zt[r * CH + (c ^ (r & SWM))];
                const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(wv[k] << 16), hi = __uint_as_float(wv[k] & 0xffff0000u);
                    q = __builtin_fmaf(lo, lo, q);
                    q = __builtin_fmaf(hi, hi, q);
                }
            }
        }
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        if (r < VQ_R && part == 0) s_e2[r] = q;
    }
    __syncthreads();
    if (t < VQ_R) {
        // |z| <= |z~| / (1 - 2^-9): inflate by 1.004 (rounding of the sum included)
        const float zzr = s_e2[t] * 1.008f, zn = sqrtf(s_e2[t]) * 1.004f, ee_max = p.en[p.n_e], en_max = p.en[p.n_e + 1];
        const float gz = 9.5367431640625e-07f * (zzr + ee_max + 2.0f * zn * en_max) + 1e-30f;
        s_e2[t] = 2.0f * (2.0f * VQ_EPS * zn * en_max + gz);
    }
    __syncthreads();
    float e2[4], smin[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        e2[rt] = s_e2[rt * 32 + c31];
        smin[rt] = INFINITY;
    }

    const int npairs = p.n_e >> 6;
    const bool uniform_rounds = npairs >= NW;      // every wave runs a first round (block-uniform condition)
    uint4 ar[4][2];
    if (w < npairs) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ar[u][0] = ld_code(p.cbp, S, 2 * w, u, lane);
            ar[u][1] = ld_code(p.cbp, S, 2 * w + 1, u, lane);
        }
    }
    // (Round 4: the B fragments requested one k step ahead of their MFMAs, two register sets - the ds_read latency then hides behind 8 MFMAs
    // instead of being waited for before every second one - measured SLOWER: 200 against 173 us for the search at M = 82 944.  The search is
    // not bound by that latency; two waves per SIMD already cover it, and the extra 16 registers cost more elsewhere.)
    for (int pp = w; pp < npairs; pp += NW) {
        f32x16 acc[2][4];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ci][rt][r] = 0.f;
        for (int s0 = 0; s0 < S; s0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u;
                const uint4 a0 = ar[u][0], a1 = ar[u][1];
                const int chunk = (2 * s + h) ^ (c31 & SWM);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const uint4 b = zt[(rt * 32 + c31) * CH + chunk];
                    acc[0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b),
                                                                        acc[0][rt], 0, 0, 0);
                    acc[1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b),
                                                                        acc[1][rt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // refill this ring slot for 4 steps ahead (rolling into this wave's next pair; past the end: a harmless
                    // reload).  Issued after the slot's last reader so the ring keeps its registers across iterations.
                    int ps = s + 4, pq = pp;
                    if (ps >= S) { ps -= S; pq += NW; }
                    if (pq >= npairs) pq = w;
                    ar[u][0] = ld_code(p.cbp, S, 2 * pq, ps, lane);
                    ar[u][1] = ld_code(p.cbp, S, 2 * pq + 1, ps, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // s = ee - 2 dot~ replaces the accumulator; running row minimum
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            const int cbase = (2 * pp + ci) * 32 + 4 * h;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 e4 = ld4(s_ee + cbase + 8 * r4);
                const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float sd = __builtin_fmaf(-2.0f, acc[ci][rt][r4 * 4 + q], ev[q]);
                        acc[ci][rt][r4 * 4 + q] = sd;
                        smin[rt] = fminf(smin[rt], sd);
                    }
            }
        }
        float thr[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            smin[rt] = fminf(smin[rt], __shfl_xor(smin[rt], 32, 64));
            if (h == 0) s_min[(rt * 32 + c31) * NW + w] = smin[rt];
        }
        // first round: wait for every wave's first minima (NW * 64 codes seen) before emitting anything -- without it each
        // wave emits against its own 64 codes only and the raw lists fill with entries the final threshold rejects
        if (pp == w && uniform_rounds) __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int row = rt * 32 + c31;
            // every wave's minimum of the row (its own included): possibly stale, always valid upper bounds of the row minimum.  One row = NW
            // consecutive floats: one per-lane address and immediates (per-wave planes needed 28 wrapped addresses, which were hoisted out of
            // the pair loop and spilled: 52 dwords of scratch per thread)
            static_assert(NW % 4 == 0, "row minima are read four at a time");
#pragma unroll
            for (int o = 0; o < NW; o += 4) {
                const f32x4_t m4 = *(lds_vfloat4 *)(s_min + row * NW + o);
                smin[rt] = fminf(fminf(smin[rt], fminf(m4[0], m4[1])), fminf(m4[2], m4[3]));
            }
            thr[rt] = smin[rt] + e2[rt];
        }
        // survivors of this pair: one bit per accumulator, one list reservation per (tile, row tile), codes only (an entry
        // emitted under a threshold that tightened later stays a candidate: a few more exact evaluations, never fewer)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                unsigned mask = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) mask |= (acc[ci][rt][r] <= thr[rt]) ? (1u << r) : 0u;
                if (mask) {
                    const int row = rt * 32 + c31;
                    unsigned slot = atomicAdd(&s_cnt[row], (unsigned)__builtin_popcount(mask));
                    while (mask) {
                        const int r = __builtin_ctz(mask);
                        mask &= mask - 1;
                        if (slot < (unsigned)VQ_CMAX)
                            s_code[row * VQ_CMAX + slot] = (unsigned short)((2 * pp + ci) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
                        ++slot;
                    }
                }
            }
    }
    __syncthreads();
    if (!FUSED) {
        // lists -> global: (M, VQ_CMAX) uint16, 8 B per thread
        for (int i = t; i < VQ_R * (VQ_CMAX / 4); i += NW * 64) {
            const int r = i / (VQ_CMAX / 4), q = i % (VQ_CMAX / 4);
            if (row0 + r < p.M)
                *reinterpret_cast<uint2 *>(p.cand + (size_t)(row0 + r) * VQ_CMAX + 4 * q) = *reinterpret_cast<const uint2 *>(s_code + r * VQ_CMAX + 4 * q);
        }
        if (t < VQ_R && row0 + t < p.M) {
            const unsigned n = s_cnt[t];
            p.cnt[row0 + t] = (unsigned short)(n > (unsigned)VQ_CMAX ? VQ_ALL : n);
        }
        return;
    }
    // ---- exact phase: 512 lanes = 64 rows x 8 candidate slots, the block's rows in two halves
    static_assert(!FUSED || NW == 8, "the fused exact phase maps 64 rows x 8 slots onto 8 waves");
    // (the thread id as an opaque value from here on: with the plain one every swizzled row address of this phase is loop-invariant for
    // the search above, gets hoisted over it and lands in scratch memory - 56 dwords per thread - because the search owns the registers)
    int te = t;
    asm volatile("" : "+v"(te));
    const int lane_e = te & 63, w_e = __builtin_amdgcn_readfirstlane(te >> 6);
    float *zf = reinterpret_cast<float *>(smem_raw);            // [64][D] fp32 over the dead bf16 image (64 * D * 4 = 128 * D * 2 bytes)
    const int D4 = D >> 2, LD4 = 31 - __builtin_clz(D4);
    for (int half = 0; half < 2; ++half) {
        const long long hrow0 = row0 + 64 * half;
        if (hrow0 >= p.M) break;                                 // (block-uniform)
        for (int i0 = te; i0 < 64 * D4; i0 += NW * 64 * 8) {      // 8 loads in flight per thread, consecutive lanes walk a row
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NW * 64, r = i >> LD4, c4 = i & (D4 - 1);
                v[u] = float4{0.f, 0.f, 0.f, 0.f};
                if (i < 64 * D4 && hrow0 + r < p.M) v[u] = ld4(p.z + (size_t)(hrow0 + r) * D + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NW * 64, r = i >> LD4, c4 = i & (D4 - 1);
                if (i < 64 * D4) *reinterpret_cast<float4 *>(zf + r * D + 4 * (c4 ^ (r & 7))) = v[u];
            }
        }
        __syncthreads();
        const int rl = te >> 3, sl = te & 7;                       // row of the half, candidate slot; a wave = 8 rows
        const long long row = hrow0 + rl;
        const bool rok = row < p.M;
        const float *zrow = zf + rl * D;
        const unsigned ncnt = rok ? s_cnt[64 * half + rl] : 0u;
        const bool all = ncnt > (unsigned)VQ_CMAX;
        const int n = all ? 0 : (int)ncnt;
        int nmax = n;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) nmax = max(nmax, __shfl_xor(nmax, sft, 64));
        float zzr = 0.f, bd = INFINITY;
        int bi = 0x7fffffff;
        for (int base = 0; base < nmax; base += 8) {
            const int k = base + sl;
            const bool act = k < n;
            int code = 0;
            float acc = 0.f, zacc = 0.f;
            if (act) {
                code = (int)s_code[(64 * half + rl) * VQ_CMAX + k];
                vq_chain_lds(zrow, rl & 7, p.cb + (size_t)code * D, D, acc, zacc);
            }
            if (base == 0) zzr = __shfl(zacc, lane_e & ~7, 64);        // slot 0 is active whenever the row has a candidate
            const float d = (zzr + s_ee[code]) - 2.0f * acc;
            if (act && (d < bd || (d == bd && code < bi))) { bd = d; bi = code; }
        }
        // rows marked "every code": the whole wave scans the codebook, 64 codes per step (constructed inputs only)
        unsigned long long allmask = __ballot(all && sl == 0);
        while (allmask) {
            const int l = __builtin_ctzll(allmask);
            allmask &= allmask - 1;
            const int ra = (w_e * 64 + l) >> 3;
            float wd = INFINITY;
            int wi = 0x7fffffff;
            for (int j0 = 0; j0 < p.n_e; j0 += 64) {
                const int code = j0 + lane_e;
                float acc = 0.f, zacc = 0.f;
                vq_chain_lds(zf + ra * D, ra & 7, p.cb + (size_t)code * D, D, acc, zacc);
                const float d = (zacc + s_ee[code]) - 2.0f * acc;
                if (d < wd) { wd = d; wi = code; }                   // ascending codes per lane_e: strict < keeps the first
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                const float od = __shfl_xor(wd, sft, 64);
                const int oi = __shfl_xor(wi, sft, 64);
                if (od < wd || (od == wd && oi < wi)) { wd = od; wi = oi; }
            }
            if ((lane_e & ~7) == l) { bd = wd; bi = wi; }              // all eight lanes of that row
        }
#pragma unroll
        for (int sft = 4; sft >= 1; sft >>= 1) {
            const float od = __shfl_xor(bd, sft, 64);
            const int oi = __shfl_xor(bi, sft, 64);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        bi = bi < 0 ? 0 : (bi >= p.n_e ? p.n_e - 1 : bi);       // rows with no finite distance keep the sentinel, as the single-pass path
        if (rok && sl == 0) p.idx[row] = (long long)bi;
        // z_q of this wave's 8 rows from the LDS copy of z, 4 rows at a time (all code rows requested before the first store)
        for (int r0 = 0; r0 < 8; r0 += 4) {
            float4 ev[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = __shfl(bi, (r0 + r) * 8, 64);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = lane_e * 4 + 256 * j;
                    ev[r][j] = float4{0.f, 0.f, 0.f, 0.f};
                    if (c < D) ev[r][j] = ld4(p.cb + (size_t)b * D + c);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = w_e * 8 + r0 + r;
                const long long rr = hrow0 + lr;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c4 = lane_e + 64 * j;
                    if (rr < p.M && 4 * c4 < D) {
                        const float4 zv = ld4(zf + lr * D + 4 * (c4 ^ (lr & 7))), ef = ev[r][j];
                        float4 q;
                        q.x = zv.x + (ef.x - zv.x);
                        q.y = zv.y + (ef.y - zv.y);
                        q.z = zv.z + (ef.z - zv.z);
                        q.w = zv.w + (ef.w - zv.w);
                        *reinterpret_cast<float4 *>(p.zq + (size_t)rr * D + 4 * c4) = q;
                    }
                }
            }
        }
        __syncthreads();                                             // the next half overwrites zf
    }
}

template <int SLOTS>
__global__ __launch_bounds__(256, 4) void vq_exact_kernel(const float *__restrict__ z, long long M, int D, const float *__restrict__ cb,
                                                       const float *__restrict__ ee, int n_e,
                                                       const unsigned short *__restrict__ cand, const unsigned short *__restrict__ cnt,
                                                       long long *__restrict__ idx, float *__restrict__ zq)
{
    constexpr int RPW = 64 / SLOTS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rl = lane / SLOTS, sl = lane % SLOTS;
    const long long wrow0 = ((long long)blockIdx.x * 4 + w) * RPW;
    if (wrow0 >= M) return;
    const long long row = wrow0 + rl;
    const bool rok = row < M;
    int n = rok ? (int)cnt[row] : 0;
    const bool all = n == VQ_ALL;
    if (all) n = 0;
    int nmax = n;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) nmax = max(nmax, __shfl_xor(nmax, s, 64));
    const float *zr = z + (size_t)(rok ? row : wrow0) * D;
    float zzr = 0.f;
    float bd = INFINITY;
    int bi = 0x7fffffff;
    for (int base = 0; base < nmax; base += SLOTS) {
        const int k = base + sl;
        const bool act = k < n;
        int code = 0;
        float acc = 0.f, zacc = 0.f;
        if (act) {
            code = (int)cand[(size_t)row * VQ_CMAX + k];
            vq_chain(zr, cb + (size_t)code * D, D, acc, zacc);
        }
        if (base == 0) zzr = __shfl(zacc, rl * SLOTS, 64);       // slot 0 is active whenever the row has a candidate
        const float d = (zzr + ee[code]) - 2.0f * acc;
        if (act && (d < bd || (d == bd && code < bi))) { bd = d; bi = code; }
    }
    // rows marked "every code": the whole wave scans the codebook, 64 codes per step
    unsigned long long allmask = __ballot(all && sl == 0);
    while (allmask) {
        const int l = __builtin_ctzll(allmask);
        allmask &= allmask - 1;
        const float *za = z + (size_t)(wrow0 + l / SLOTS) * D;
        float wd = INFINITY;
        int wi = 0x7fffffff;
        for (int j0 = 0; j0 < n_e; j0 += 64) {
            const int code = j0 + lane;
            float acc = 0.f, zacc = 0.f;
            vq_chain(za, cb + (size_t)code * D, D, acc, zacc);
            const float d = (zacc + ee[code]) - 2.0f * acc;
            if (d < wd) { wd = d; wi = code; }                   // ascending codes per lane: strict < keeps the first
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const float od = __shfl_xor(wd, s, 64);
            const int oi = __shfl_xor(wi, s, 64);
            if (od < wd || (od == wd && oi < wi)) { wd = od; wi = oi; }
        }
        if (lane == l) { bd = wd; bi = wi; }
    }
#pragma unroll
    for (int s = SLOTS / 2; s >= 1; s >>= 1) {
        const float od = __shfl_xor(bd, s, 64);
        const int oi = __shfl_xor(bi, s, 64);
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    bi = bi < 0 ? 0 : (bi >= n_e ? n_e - 1 : bi);       // rows with no finite distance keep the sentinel, as the single-pass path
    if (rok && sl == 0) idx[row] = (long long)bi;
    // z_q rows, 4 at a time (loads of all four in flight before the first store)
    for (int r0 = 0; r0 < RPW; r0 += 4) {
        float4 zv[4][2], ev[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = __shfl(bi, (r0 + r) * SLOTS, 64);
            const long long rr = wrow0 + r0 + r;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = lane * 4 + 256 * j;
                if (rr < M && c < D) {
                    zv[r][j] = ld4(z + (size_t)rr * D + c);
                    ev[r][j] = ld4(cb + (size_t)b * D + c);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long rr = wrow0 + r0 + r;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = lane * 4 + 256 * j;
                if (rr < M && c < D) {
                    const float4 zf = zv[r][j], ef = ev[r][j];
                    float4 q;
                    q.x = zf.x + (ef.x - zf.x);
                    q.y = zf.y + (ef.y - zf.y);
                    q.z = zf.z + (ef.z - zf.z);
                    q.w = zf.w + (ef.w - zf.w);
                    *reinterpret_cast<float4 *>(zq + (size_t)rr * D + c) = q;
                }
            }
        }
    }
}

constexpr int VQ_NW = 8;

size_t cand_lds_bytes(int n_e, int D)
{
    return (size_t)VQ_R * D * 2 + (size_t)n_e * 4 + (size_t)VQ_NW * VQ_R * 4 + VQ_R * 8 + (size_t)VQ_R * VQ_CMAX * 2;
}

unsigned long long g_cand_attr_devs[2] = {0ull, 0ull};      // [list-only | one-launch] instantiation: bit = device

int exact_slots()
{
    static const int v = [] { const char *e = getenv("FEMASR_VQ_SLOTS"); return (e && atoi(e) == 4) ? 4 : 8; }();      // (thread-safe one-time init)
    return v;
}

}  // namespace

extern "C" {

int femasr_vq_twopass_ok(int n_e, int D)
{
    return (D >= 64 && D <= 512 && (D & (D - 1)) == 0 && n_e >= 64 && n_e <= 1024 && n_e % 64 == 0) ? 1 : 0;
}

size_t femasr_vq_aux_bytes(int n_e, int D)
{
    if (!femasr_vq_twopass_ok(n_e, D)) return 0;
    return (size_t)n_e * D * 2 + (size_t)n_e * sizeof(float) + 2 * sizeof(float);
}

// aux = [packed bf16 codebook | en[n_e] | max ee, max en]
int femasr_vq_prepare(void *stream, const float *cb, const float *ee, int n_e, int D, void *aux)
{
    FEMASR_REQUIRE(cb && ee && aux, "vq_prepare: null pointer");
    FEMASR_REQUIRE(femasr_vq_twopass_ok(n_e, D), "vq_prepare: shape (n_e=%d, e_dim=%d) not supported by the two-pass search", n_e, D);
    hipStream_t s = (hipStream_t)stream;
    float *en = (float *)((char *)aux + (size_t)n_e * D * 2);
    hipLaunchKernelGGL(vq_pack_codebook_kernel, dim3(256), dim3(256), 0, s, cb, ee, n_e, D, (uint4 *)aux, en);
    FEMASR_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(vq_codebook_max_kernel, dim3(1), dim3(64), 0, s, ee, n_e, en);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

size_t femasr_vq_scratch_bytes(int64_t M, int n_e)
{
    const size_t m64 = (size_t)((M + 63) / 64) * 64;
    const size_t gemm = ((size_t)M * (size_t)(n_e / 128 > 0 ? n_e / 128 : 1) * 2 + m64 + 64) * sizeof(float);
    const size_t two = m64 * VQ_CMAX * 2 + m64 * 2 + 256;
    return gemm > two ? gemm : two;
}

static int launch_candidates(hipStream_t s, const VqCandParams &p, bool fused)
{
    const int lds = (int)cand_lds_bytes(p.n_e, p.D);
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    unsigned long long *mask = &g_cand_attr_devs[fused ? 1 : 0];
    if (!(__atomic_load_n(mask, __ATOMIC_ACQUIRE) & bit)) {
        FEMASR_CHECK_HIP(hipFuncSetAttribute(fused ? (const void *)vq_candidates_kernel<VQ_NW, true> : (const void *)vq_candidates_kernel<VQ_NW, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        __atomic_fetch_or(mask, bit, __ATOMIC_RELEASE);
    }
    const dim3 grid((unsigned)((p.M + VQ_R - 1) / VQ_R)), block(VQ_NW * 64);
    if (fused) hipLaunchKernelGGL((vq_candidates_kernel<VQ_NW, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((vq_candidates_kernel<VQ_NW, false>), grid, block, lds, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

// pass 1 alone (tests: the exact argmin must be among the candidates).  cand (M, VQ_CMAX = 32) u16, cnt (M) u16 with
// 0xFFFF = "every code".
int femasr_vq_candidates(void *stream, const float *z, int64_t M, int D, const void *aux, const float *ee, int n_e,
                         uint16_t *cand, uint16_t *cnt)
{
    FEMASR_REQUIRE(z && aux && ee && cand && cnt && M > 0, "vq_candidates: bad args");
    FEMASR_REQUIRE(femasr_vq_twopass_ok(n_e, D), "vq_candidates: shape (n_e=%d, e_dim=%d) not supported", n_e, D);
    const float *en = (const float *)((const char *)aux + (size_t)n_e * D * 2);
    VqCandParams p{z, (long long)M, D, n_e, (const uint4 *)aux, ee, en, cand, cnt, nullptr, nullptr, nullptr};
    return launch_candidates((hipStream_t)stream, p, false);
}

// The lookup.  Default: ONE launch (candidate search + exact phase in the same block, FUSED above); FEMASR_VQ_FUSED=0 keeps the round-2
// form - the candidate lists through global memory to vq_exact_kernel - for A/B.  Same results bit for bit.
int femasr_vq_twopass(void *stream, const float *z, int64_t M, int D, const float *cb, const void *aux, const float *ee, int n_e,
                      int64_t *idx, float *zq, void *scratch)
{
    FEMASR_REQUIRE(z && cb && aux && ee && idx && zq && scratch && M > 0, "vq_twopass: bad args");
    FEMASR_REQUIRE(M < (1ll << 31) - 256, "vq: too many rows");
    FEMASR_REQUIRE(femasr_vq_twopass_ok(n_e, D), "vq_twopass: shape (n_e=%d, e_dim=%d) not supported", n_e, D);
    hipStream_t s = (hipStream_t)stream;
    static const bool fused = [] { const char *e = getenv("FEMASR_VQ_FUSED"); return !(e && atoi(e) == 0); }();      // (thread-safe one-time init)
    const float *en = (const float *)((const char *)aux + (size_t)n_e * D * 2);
    if (fused) {
        VqCandParams p{z, (long long)M, D, n_e, (const uint4 *)aux, ee, en, nullptr, nullptr, cb, (long long *)idx, zq};
        return launch_candidates(s, p, true);
    }
    const size_t m64 = (size_t)((M + 63) / 64) * 64;
    uint16_t *cand = (uint16_t *)scratch;
    uint16_t *cnt = cand + m64 * VQ_CMAX;
    const int rc = femasr_vq_candidates(stream, z, M, D, aux, ee, n_e, cand, cnt);
    if (rc) return rc;
    if (exact_slots() == 8)
        hipLaunchKernelGGL(vq_exact_kernel<8>, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, s, z, (long long)M, D, cb, ee, n_e, cand, cnt,
                           (long long *)idx, zq);
    else
        hipLaunchKernelGGL(vq_exact_kernel<4>, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, s, z, (long long)M, D, cb, ee, n_e, cand, cnt,
                           (long long *)idx, zq);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

}  // extern "C"
