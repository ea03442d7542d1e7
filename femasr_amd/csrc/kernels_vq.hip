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
#include <type_traits>

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int VQ_R = 128;                  // rows per block of pass 1 (4 MFMA column tiles)
constexpr int VQ_CMAX = 32;                // candidate slots per row (LDS list of pass 1 = the list handed to pass 2)
constexpr int VQ_ALL = 0xFFFF;             // count marker: evaluate every code
constexpr float VQ_EPS = 0.0042f;
// one-launch form: line staging of the exact phase - per wave two buffers of (8 candidate slots + the rows of z) x 8 rows x 128 B
constexpr int VQ_STAGE_BUF = 9 * 1024;
constexpr int VQ_STAGE_BYTES = 8 * 2 * VQ_STAGE_BUF;              // 147 456: covers the bf16 image of the search (<= 131 072)
__host__ __device__ constexpr size_t vq_fused_list0(int D) { return (size_t)VQ_R * D * 2 > (size_t)VQ_STAGE_BYTES ? (size_t)VQ_R * D * 2 : (size_t)VQ_STAGE_BYTES; }

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

// pass 1.  Block = NW waves (2 per SIMD: one wave's bound arithmetic overlaps the other's MFMAs), 128 rows of z as a bf16
// image in LDS (the MFMA B operand); wave w takes code-tile pairs w, w+NW, ..: per 16-deep k step 2 A fragments from global
// (L2-resident packed codebook, 4-step register ring), 4 B fragments from LDS, 8 MFMAs.
// Bound per row (see the header): every code's |d - zz - s| <= Erow, s = ee - 2 dot~, Erow = 2 VQ_EPS |z| |e|max + gz;
// a code survives iff s <= min_j s_j + 2 Erow.
// FUSED (round 4, the default): the block also runs the EXACT phase for its 128 rows - the lists never leave LDS, idx and z_q are written
// here, ONE launch.  The bf16 image is dead after the search; round 5: its LDS becomes per-wave staging buffers through which the candidates'
// code rows and the rows of z arrive line-coalesced (see the exact phase below), z_q = z + (e - z) is formed from registers.
template <int NW, bool FUSED>
__global__ __launch_bounds__(NW * 64, NW / 4) void vq_candidates_kernel(const VqCandParams p)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, c31 = lane & 31, h = lane >> 5;
    const int D = p.D, CH = D >> 3, S = D >> 4, SWM = (CH < 16 ? CH : 16) - 1, LCH = 31 - __builtin_clz(CH);
    uint4 *zt = reinterpret_cast<uint4 *>(smem_raw);                               // [128][CH] 16-B chunks, XOR-swizzled
    // FUSED: the exact phase's line staging ring (8 waves x 2 x 9 KiB) takes the first 144 KiB once the image is dead; what it needs of the
    // lists (ee, counts, codes) lies behind that, what only the search needs (running minima, the row bounds) inside its last 16 KiB
    const size_t list0 = FUSED ? vq_fused_list0(D) : (size_t)VQ_R * D * 2;
    float *s_ee = reinterpret_cast<float *>(smem_raw + list0);
    // (an explicit LDS pointer: through a generic `volatile float *` the address space is not inferred and every access becomes a flat one
    // with a 64-bit address)
    typedef __attribute__((address_space(3))) volatile float lds_vfloat;
    typedef __attribute__((address_space(3))) const volatile f32x4_t lds_vfloat4;
    float *search_only = FUSED ? reinterpret_cast<float *>(smem_raw + VQ_STAGE_BYTES - (NW * VQ_R + VQ_R) * 4) : s_ee + p.n_e;
    lds_vfloat *s_min = (lds_vfloat *)search_only;                                 // [128][NW] running min of s, read without barriers
    float *s_e2 = search_only + NW * VQ_R;                                         // [128] |z~|^2, then 2 Erow
    unsigned *s_cnt = reinterpret_cast<unsigned *>(FUSED ? s_ee + p.n_e : s_e2 + VQ_R);
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
    // ---- exact phase: 512 lanes = 64 rows x 8 candidate slots, the block's rows in two halves (the thread id as an opaque value from here on:
    // with the plain one the addresses of this phase are loop-invariant for the search above and get hoisted over it)
    static_assert(!FUSED || NW == 8, "the fused exact phase maps 64 rows x 8 slots onto 8 waves");
    int te = t;
    asm volatile("" : "+v"(te));
    const int lane_e = te & 63, w_e = __builtin_amdgcn_readfirstlane(te >> 6);
    // Round 5: the code rows arrive LINE-COALESCED.  A wave = 8 rows x 8 candidate slots, as before, but a 128-byte line of a candidate's
    // code row is fetched by the 8 lanes of its ROW group (16 bytes each, one load instruction per slot: 8 full lines per instruction
    // instead of 64 lanes x 16 bytes of 64 different lines), written to the wave's staging buffer as [slot][row][chunk], and read back by
    // the owner lane (row, slot) with 8 ds_read_b128.  The chunk a lane fetches is xor-ed with the slot so that the 16 lanes of a read
    // group (2 rows x 8 slots) hit 16 different bank quads.  The row's line of z comes the same way (one more load: 8 rows x 128 B, read
    // back as a broadcast by the row's slots) and STAYS in the fetching lane's registers: after the last line the 8 lanes of a row hold the
    // row between them (16 lines x 16 B each) and form z_q = z + (e - z) from it, again 8 lanes per line - no fp32 copy of the rows in LDS,
    // z is read twice per launch (search, here) as before.  No barriers: a staging buffer is private to its wave.
    // Operation order of the chains = vq_chain's (one 32-float burst per line).
    // (The same fetch through LDS-DMA - global_load_lds_dwordx4, no registers, two buffers per wave - was built first and is bit-identical:
    // 161 us for this phase against the 143 of round 4; the copy path tops out near 6.4 TB/s for the chip (MI355X_MICROARCH.md
    // "ldsdma-fill") and the phase moves 650 MB through it.)
    unsigned char *stage_gen = smem_raw + (size_t)w_e * (2 * VQ_STAGE_BUF);
    const int rl8 = lane_e >> 3, sl = lane_e & 7;
    // (the number of lines is a COMPILE-TIME constant of the body: with a run-time `c < nline` guard in the unrolled line loops the register
    // allocator put the kept rows of z - 64 registers - into scratch memory)
    auto exact_phase = [&](auto nline_c) {
    constexpr int nline = decltype(nline_c)::value;              // D / 32: 2, 4, 8 or 16
    for (int half = 0; half < 2; ++half) {
        const long long hrow0 = row0 + 64 * half;
        if (hrow0 >= p.M) break;                                 // (block-uniform)
        const int lrow = 64 * half + w_e * 8 + rl8;              // row of the block; a wave = 8 rows x 8 slots
        const long long row = row0 + lrow;
        const bool rok = row < p.M;
        const unsigned ncnt = rok ? s_cnt[lrow] : 0u;
        const bool all = ncnt > (unsigned)VQ_CMAX;
        const int n = all ? 0 : (int)ncnt;
        int nmax = n;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) nmax = max(nmax, __shfl_xor(nmax, sft, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        const int zk = rl8 >> 1;                                 // (z chunks xor-ed with the row pair: the four rows of a read group in four bank quads)
        const int zoff = 4 * (sl ^ zk);                          // this lane's chunk of every line of its row (z, z_q and the winner's code row)
        const float *zsrc = p.z + (size_t)(rok ? row : p.M - 1) * D + zoff;
        float4 zkeep[16];                                        // (statically indexed: the line loops below are fully unrolled)
        float bd = INFINITY;
        int bi = 0x7fffffff;
        // rows marked "every code": the whole wave scans the codebook, 64 codes per step (constructed inputs only).
        // First: the passes below keep the wave's rows of z in registers until z_q is written - nothing that needs many registers may sit between
        unsigned long long allmask = __ballot(all && sl == 0);
        while (allmask) {
            const int l = __builtin_ctzll(allmask);
            allmask &= allmask - 1;
            const long long ra_ = hrow0 + ((w_e * 64 + l) >> 3);
            float wd = INFINITY;
            int wi = 0x7fffffff;
            for (int j0 = 0; j0 < p.n_e; j0 += 64) {
                const int code = j0 + lane_e;
                float acc = 0.f, zacc = 0.f;
                vq_chain(p.z + (size_t)ra_ * D, p.cb + (size_t)code * D, D, acc, zacc);
                const float d = (zacc + s_ee[code]) - 2.0f * acc;
                if (d < wd) { wd = d; wi = code; }                   // ascending codes per lane: strict < keeps the first
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                const float od = __shfl_xor(wd, sft, 64);
                const int oi = __shfl_xor(wi, sft, 64);
                if (od < wd || (od == wd && oi < wi)) { wd = od; wi = oi; }
            }
            if ((lane_e & ~7) == l) { bd = wd; bi = wi; }              // all eight lanes of that row
        }
        auto run = [&](int base, auto keep_c) {                // one pass over the rows' slots base .. base + 7; keep_c: the pass that keeps z
            constexpr bool KEEP = decltype(keep_c)::value;
            const int nb = nmax - base < 8 ? nmax - base : 8;    // slots in use by any row of the wave (uniform)
            // the eight codes of this lane's row at slots base .. base + 7 (entries past the row's count are never fetched)
            const uint4 cw = *reinterpret_cast<const uint4 *>(s_code + lrow * VQ_CMAX + base);
            const unsigned cws[4] = {cw.x, cw.y, cw.z, cw.w};
            auto code_of = [&](int i) -> int { return (int)((cws[i >> 1] >> (16 * (i & 1))) & 0xffffu); };
            unsigned eoff[8];                                    // float offsets into the codebook (< 2^19): one register per slot
#pragma unroll
            for (int i = 0; i < 8; ++i) eoff[i] = (unsigned)((base + i < n ? code_of(i) : 0) * D + 4 * (sl ^ i));
            const bool act = base + sl < n;
            int code = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) code = sl == i ? code_of(i) : code;
            if (!act) code = 0;
            float acc = 0.f, zacc = 0.f;
            auto load_line = [&](int c, float4 (&r)[8], float4 &zr) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    r[i] = float4{0.f, 0.f, 0.f, 0.f};
                    if (i < nb && base + i < n) r[i] = ld4(p.cb + (eoff[i] + 32u * c));
                }
                zr = ld4(zsrc + 32 * c);
            };
            auto put_line = [&](const float4 (&r)[8], const float4 &zr, int buf) {
                float4 *dst = reinterpret_cast<float4 *>(stage_gen + buf * VQ_STAGE_BUF) + lane_e;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < nb && base + i < n) dst[64 * i] = r[i];
                dst[512] = zr;
            };
            auto chain_line = [&](int buf) {      // (LDS operations of one wave complete in order: these reads see put_line's stores)
                if (!act) return;
                const float4 *eb = reinterpret_cast<const float4 *>(stage_gen + buf * VQ_STAGE_BUF + sl * 1024 + rl8 * 128);
                const float4 *zb = reinterpret_cast<const float4 *>(stage_gen + buf * VQ_STAGE_BUF + 8192 + rl8 * 128);
#pragma unroll
                for (int g = 0; g < 4; ++g) {     // vq_chain_burst's order, 8 channels at a time (16 live registers instead of 64)
                    const float4 z0 = zb[(2 * g) ^ zk], z1 = zb[(2 * g + 1) ^ zk], e0 = eb[(2 * g) ^ sl], e1 = eb[(2 * g + 1) ^ sl];
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
            };
            // one register set: a line's registers are free once it sits in LDS, so line c + 1 is requested before line c is evaluated
            float4 ra[8], zt2[2];                                // (zt2: the row's line in the passes that do not keep it)
            load_line(0, ra, KEEP ? zkeep[0] : zt2[0]);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c < nline) {                                 // (uniform)
                    put_line(ra, KEEP ? zkeep[c] : zt2[c & 1], c & 1);
                    if (c + 1 < 16 && c + 1 < nline) load_line(c + 1, ra, KEEP ? zkeep[c + 1] : zt2[(c + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    chain_line(c & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const float d = (zacc + s_ee[code]) - 2.0f * acc;    // every active lane of a row has the row's |z|^2 chain (same sequence)
            if (act && (d < bd || (d == bd && code < bi))) { bd = d; bi = code; }
        };
        run(0, std::true_type{});                              // (also when no row of the wave has a list: it fetches the rows of z for z_q)
        for (int base = 8; base < nmax; base += 8) run(base, std::false_type{});      // rows with more than 8 candidates (~1 % of them)
#pragma unroll
        for (int sft = 4; sft >= 1; sft >>= 1) {
            const float od = __shfl_xor(bd, sft, 64);
            const int oi = __shfl_xor(bi, sft, 64);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        bi = bi < 0 ? 0 : (bi >= p.n_e ? p.n_e - 1 : bi);       // rows with no finite distance keep the sentinel, as the single-pass path
        if (rok && sl == 0) p.idx[row] = (long long)bi;
        // z_q = z + (e - z): the row's 8 lanes hold z between them; the winner's code row and the stores go 8 lanes per 128-byte line too
        // (every line of the code row requested before the first store)
        {
            const float *ewin = p.cb + (size_t)bi * D + zoff;
            float *qdst = p.zq + (size_t)(rok ? row : 0) * D + zoff;
            float4 ev[16];
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < nline) ev[c] = ld4(ewin + 32 * c);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c < nline && rok) {
                    const float4 z4 = zkeep[c], ef = ev[c];
                    float4 q;
                    q.x = z4.x + (ef.x - z4.x);
                    q.y = z4.y + (ef.y - z4.y);
                    q.z = z4.z + (ef.z - z4.z);
                    q.w = z4.w + (ef.w - z4.w);
                    *reinterpret_cast<float4 *>(qdst + 32 * c) = q;
                }
            }
        }
    }
    };
    switch (D >> 5) {                                            // (femasr_vq_twopass_ok: e_dim in {64, 128, 256, 512})
    case 16: exact_phase(std::integral_constant<int, 16>{}); break;
    case 8: exact_phase(std::integral_constant<int, 8>{}); break;
    case 4: exact_phase(std::integral_constant<int, 4>{}); break;
    default: exact_phase(std::integral_constant<int, 2>{}); break;
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

size_t cand_lds_bytes(int n_e, int D, bool fused)
{
    if (fused) return vq_fused_list0(D) + (size_t)n_e * 4 + VQ_R * 4 + (size_t)VQ_R * VQ_CMAX * 2;      // 160 256 at n_e = 1024
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
    const int lds = (int)cand_lds_bytes(p.n_e, p.D, fused);
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
