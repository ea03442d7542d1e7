// kernels_wino.hip — 3x3 stride-1 pad-1 convs in the Winograd F(2x2, 3x3) form on the fp32 matrix cores.
//
// Used for the convs BEHIND the codebook lookup of single-codebook networks (after_quant, decoder ResBlocks, the LQ
// encoder's up-blocks that only make skip features; fema_utils.py:65-84, femasr_arch.py:196-207,298): they cannot move a VQ
// index, and the form needs 16 multiplies per 2x2 outputs where the direct sweep needs 36 (2.25x fewer MFMAs).  All fp32;
// the order of every addition is the one of oracle/femasr_oracle.c orc_conv3x3_winograd, so the result is bit-identical to
// that restatement (and within fp32 rounding, ~1e-6 relative, of the direct form).
//
// Block = 8 waves, one 8 x 16 output tile (= 4 x 8 Winograd tiles = ONE 32-row MFMA tile) x BN channels:
//   per 32-channel block of the input:
//     T phase  every thread transforms two (Winograd tile, channel) 4x4 input patches from the staged halo patch
//              (GN + SiLU already applied while staging, zeros outside the image) into V[16][32 tiles][32 ch] in LDS
//     M phase  wave w owns frequency components 2w, 2w+1: M_k[32 tiles][BN] += V_k[32][32] . U_k[32][BN]
//              (v_mfma_f32_32x32x2_f32, A from LDS, U fragments from L2 in the fragment-major layout of a 4x4-tap conv);
//              the next channel block's patch is fetched and staged underneath
//   epilogue   accumulators -> LDS by component, one wave per (32-pixel block, 32-channel tile) applies A^T M A and lands
//              on exactly the register layout of the direct halo kernel (element r of a lane = the same pixel), so bias /
//              residual adds / stores / the fused GroupNorm partial moments are the same code in the same order.
#include "conv_common.h"
#include "detmath.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef FEMASR_WINO_TT      // tools/build_debug.sh: per-wave cycle shares of the kernel's phases
__device__ unsigned long long g_wi_tt[8];
#define WTT(slot) { const unsigned long long now_ = __builtin_readcyclecounter(); tt_acc[slot] += now_ - tt_last; tt_last = now_; }
#define WTT_INIT unsigned long long tt_acc[7] = {0, 0, 0, 0, 0, 0, 0}; unsigned long long tt_last = __builtin_readcyclecounter(); const unsigned long long tt_first = tt_last;
#define WTT_END { if (lane == 0) { for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_wi_tt[i_], tt_acc[i_]); atomicAdd(&g_wi_tt[7], __builtin_readcyclecounter() - tt_first); } }
#else
#define WTT(slot) {}
#define WTT_INIT
#define WTT_END {}
#endif

namespace {

constexpr int WI_PW = 18, WI_PP = 180;                         // halo patch 10 x 18 pixels
constexpr int WI_NT = 512;
constexpr int WI_PUNITS = (WI_PP * 8 + WI_NT - 1) / WI_NT;     // 3
constexpr int WI_PROWS = WI_NT / 8;                            // 64
constexpr int WI_PSZ = (((WI_PP + 1) * ALD + 3) / 4) * 4;      // floats per patch buffer (pixel 180 = write-only dummy)
constexpr int WI_VSZ = 16 * 32 * ALD;                          // floats of V / of one epilogue slab
constexpr int WI_RED = 4 * 64 * 2 * 2;                         // floats: [4 q][<= 64 groups][2] doubles

template <int TNW>
constexpr size_t wino_lds_bytes()
{
    return (size_t)(2 * WI_VSZ + WI_RED) * sizeof(float);      // epilogue: 2 slabs + moments; main loop: 2 patches + V (smaller)
}
static_assert(2 * WI_PSZ + WI_VSZ <= 2 * WI_VSZ, "main-loop buffers fit under the epilogue slabs");

template <int TNW, int PRO>          // TNW: 32-channel tiles per wave and component (BN = 32 TNW)
__global__ __launch_bounds__(WI_NT, 2) void conv3x3_wino_kernel(const ConvParams p)
{
    constexpr int BN = 32 * TNW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                      // [2][WI_PSZ]
    float *Vs = smem + 2 * WI_PSZ;         // [16][32][ALD]

    const int t = threadIdx.x, lane = t & 63;
    WTT_INIT
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c31 = lane & 31, hh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB;
    int tile = L / p.NB;
    const int tx = tile % p.tilesX;
    tile /= p.tilesX;
    const int ty = tile % p.tilesY;
    const int n = tile / p.tilesY;
    const int oy0 = ty * 8, ox0 = tx * 16, n0 = nb * BN;
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;

    // ---- patch staging units: (pixel = (t>>3) + 64 i, channel quad kq)
    const int kq = t & 7;
    unsigned poff[WI_PUNITS];
    unsigned pmask = 0;
#pragma unroll
    for (int i = 0; i < WI_PUNITS; ++i) {
        const int pix = (t >> 3) + WI_PROWS * i;
        const int ppy = pix / WI_PW, ppx = pix - ppy * WI_PW;
        const int sy = sy0 + ppy, sx = sx0 + ppx;
        const bool ok = (pix < WI_PP) & (sy >= 0) & (sy < p.H) & (sx >= 0) & (sx < p.W);
        poff[i] = ok ? (unsigned)((((size_t)n * p.H + sy) * p.W + sx) * p.Cin + 4 * kq) : 0u;
        pmask |= (ok ? 1u : 0u) << i;
    }
    float4 rp[WI_PUNITS], ga, gb;
    auto load_patch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < WI_PUNITS; ++i) rp[i] = ld4(p.in + (size_t)poff[i] + (size_t)cc * BK);
        if (PRO == FEMASR_PRO_GN_SILU) {
            ga = ld4(p.pro_a + (size_t)n * p.Cin + cc * BK + 4 * kq);
            gb = ld4(p.pro_b + (size_t)n * p.Cin + cc * BK + 4 * kq);
        }
    };
    auto store_patch = [&](int buf) {
        float *Pb = Ps + buf * WI_PSZ;
#pragma unroll
        for (int i = 0; i < WI_PUNITS; ++i) {
            const int pix = (t >> 3) + WI_PROWS * i;
            if (i == WI_PUNITS - 1 && pix >= WI_PP) continue;
            float4 v = rp[i];
            if (PRO == FEMASR_PRO_GN_SILU) {
                const det_f32x2 s0 = det_silu2(__builtin_elementwise_fma(det_f32x2{v.x, v.y}, det_f32x2{ga.x, ga.y}, det_f32x2{gb.x, gb.y}));
                const det_f32x2 s1 = det_silu2(__builtin_elementwise_fma(det_f32x2{v.z, v.w}, det_f32x2{ga.z, ga.w}, det_f32x2{gb.z, gb.w}));
                v = make_float4(s0[0], s0[1], s1[0], s1[1]);
            }
            if (!(pmask & (1u << i))) v = make_float4(0.f, 0.f, 0.f, 0.f);     // zero padding AFTER the activation
            float *dst = Pb + pix * ALD + 4 * kq;
            dst[0] = v.x;
            dst[1] = v.y;
            dst[2] = v.z;
            dst[3] = v.w;
        }
    };

    // ---- input transform items: channel t&31, Winograd tiles (t>>5) and (t>>5) + 16
    const int tci = t & 31;
    auto transform = [&](const float *Pb) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int wt = (t >> 5) + 16 * u, wr = wt >> 3, wc = wt & 7;
            const float *src = Pb + ((2 * wr) * WI_PW + 2 * wc) * ALD + tci;
            float d[4][4], tt[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = src[(i * WI_PW + j) * ALD];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt[0][j] = d[0][j] - d[2][j];
                tt[1][j] = d[1][j] + d[2][j];
                tt[2][j] = d[2][j] - d[1][j];
                tt[3][j] = d[1][j] - d[3][j];
            }
            float *dst = Vs + wt * ALD + tci;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dst[(i * 4 + 0) * 32 * ALD] = tt[i][0] - tt[i][2];
                dst[(i * 4 + 1) * 32 * ALD] = tt[i][1] + tt[i][2];
                dst[(i * 4 + 2) * 32 * ALD] = tt[i][2] - tt[i][1];
                dst[(i * 4 + 3) * 32 * ALD] = tt[i][1] - tt[i][3];
            }
        }
    };

    f32x16 acc[2][TNW];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

    // U fragments: [chunk q = cb*16 + component][32-column tile][16-byte group g][lane][4 k-pairs]: one wave load = 1 KiB of
    // consecutive bytes (8 full cache lines).  Address = uniform pointer + lane * 16 bytes: no per-load VALU arithmetic.
    const size_t wstride = (size_t)p.NT32 << 10;
    int wt_[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) wt_[j] = wtile(n0, j, p.NT32);
    const unsigned lw = (unsigned)lane * 16u;
    const int k0 = 2 * wave;
    auto ldw = [&](size_t q, int g, int j) -> f32x4_t { return ldg4_u32(p.w_wino + q * wstride + ((size_t)wt_[j] * 4 + g) * 256, lw); };

    const int ncc = p.Cin / BK;
    load_patch(0);
    f32x4_t bq[2][TNW];          // fragment groups ping-pong between the two sets (8 steps per channel block: parity is stable)
#pragma unroll
    for (int j = 0; j < TNW; ++j) bq[0][j] = ldw((size_t)k0, 0, j);
    store_patch(0);
    __syncthreads();

    // ---- output-stage geometry (needed early: the residual of the first round is fetched under the last MFMA phase)
    const bool gnp = p.gn_part != nullptr;
    const int cg = p.Cout >> 5, gpb = gnp ? BN / cg : 1;
    const float *ra = p.res1 ? p.res1 : p.res2, *rb = (p.res1 && p.res2) ? p.res2 : nullptr;
    const size_t obase = (((size_t)n * p.Ho + oy0) * p.Wo + ox0) * p.Cout + n0;
    const int q = wave & 3, jj = wave >> 2;            // this wave's 32-pixel block and slab in the output stage
    // element r of (row tile q, column tile j): pixel row 2q + (r>>3), column (r&3) + 8((r>>2)&1) + 4 hh, channel 32 j + c31.
    // Address = uniform part (SGPRs: q, j come from the wave index) + one per-lane offset, as in conv3x3_halo_kernel.
    const unsigned loff = (unsigned)(4 * hh) * (unsigned)p.Cout + (unsigned)c31;
    const bool full = (oy0 + 8 <= p.Ho) && (ox0 + 16 <= p.Wo) && (n0 + BN <= p.Cout);
    auto uoff = [&](int j, int r) -> size_t {
        return obase + (size_t)((2 * q + (r >> 3)) * p.Wo + (r & 3) + 8 * ((r >> 2) & 1)) * p.Cout + j * 32;
    };
    auto ok_u = [&](int j, int r) -> bool {
        return full || ((oy0 + 2 * q + (r >> 3)) < p.Ho && (ox0 + (r & 3) + 8 * ((r >> 2) & 1)) < p.Wo && (n0 + j * 32) < p.Cout);
    };
    auto ok_l = [&](int j, int r) -> bool {
        return full || ((oy0 + 2 * q + (r >> 3)) < p.Ho && (ox0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * hh) < p.Wo && (n0 + j * 32 + c31) < p.Cout);
    };
    float rv[16];
    auto fetch_res = [&](const float *src, int j, float (&dst)[16]) {     // branch-free batch; masked elements read the tile origin
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = ldg_u32(src + (ok_u(j, r) ? uoff(j, r) : obase), ok_l(j, r) ? 4u * loff : 0u);
    };

    const float *Va = Vs + c31 * ALD + hh;           // A fragment of component k, k-pair kk: Va[k*32*ALD + 2*kk]
    WTT(0)
    for (int cc = 0; cc < ncc; ++cc) {
        transform(Ps + (cc & 1) * WI_PSZ);
        WTT(1)
        __syncthreads();                               // V complete; the patch buffer cc&1 is free again
        WTT(2)
        const int ccn = cc + 1 < ncc ? cc + 1 : cc;
        const bool more = cc + 1 < ncc;
        if (more) load_patch(ccn);
        float aq[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) aq[0][e] = Va[k0 * 32 * ALD + 2 * e];
#pragma unroll
        for (int s = 0; s < 8; ++s) {                  // s = component * 4 + 16-byte group of k-pairs
            const int comp = s >> 2, cur = s & 1, nxt = cur ^ 1;
            {   // next fragment group: this block's next group, or the first group of the next channel block
                const int sn = s + 1;
                const size_t qn = sn < 8 ? (size_t)(cc * 16 + k0 + (sn >> 2)) : (size_t)(ccn * 16 + k0);
                const int gn = sn < 8 ? (sn & 3) : 0;
#pragma unroll
                for (int j = 0; j < TNW; ++j) bq[nxt][j] = ldw(qn, gn, j);
                if (sn < 8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) aq[nxt][e] = Va[(k0 + (sn >> 2)) * 32 * ALD + 8 * (sn & 3) + 2 * e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
                    acc[comp][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][e], bq[cur][j][e], acc[comp][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 3 && more) store_patch((cc + 1) & 1);      // (uniform) the loads were issued 64 MFMAs ago
        }
        WTT(3)
        __syncthreads();                               // every wave is done with V; the next patch is staged
        WTT(4)
    }

    // ---------------------------------------------------------------------------------------------------------------
    // epilogue.  Slabs Ms[jj][k][wt][c] (two 32-channel tiles per round) overlay the patch / V buffers.
    float *Ms = smem;
    double *red = reinterpret_cast<double *>(smem + 2 * WI_VSZ);
    constexpr int ROUNDS = (TNW + 1) / 2;
    if (ra && jj < TNW) fetch_res(ra, jj, rv);         // the first round's residual flies under the slab exchange
#pragma unroll
    for (int rnd = 0; rnd < ROUNDS; ++rnd) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int sj = 0; sj < 2; ++sj) {
                if (2 * rnd + sj >= TNW) continue;
                float *dst = Ms + ((size_t)(sj * 16 + k0 + c) * 32) * ALD + c31;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * hh) * ALD] = acc[c][2 * rnd + sj][r];
            }
        __syncthreads();
        WTT(5)
        const int j = 2 * rnd + jj;                    // 32-channel tile of the block handled by this wave now
        if (j < TNW) {
            float rn[16], r2[16];
            const bool next = rnd + 1 < ROUNDS && j + 2 < TNW;
            if (ra && next) fetch_res(ra, j + 2, rn);  // the next round's residual flies under this round's arithmetic
            if (rb) fetch_res(rb, j, r2);
            float v[16];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int wt = q * 8 + (t4 >> 1) * 4 + 2 * hh + (t4 & 1);
                const float *src = Ms + ((size_t)(jj * 16) * 32 + wt) * ALD + c31;
                float m[16], sm[2][4];
#pragma unroll
                for (int k = 0; k < 16; ++k) m[k] = src[(size_t)k * 32 * ALD];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    sm[0][x] = (m[0 * 4 + x] + m[1 * 4 + x]) + m[2 * 4 + x];
                    sm[1][x] = (m[1 * 4 + x] - m[2 * 4 + x]) - m[3 * 4 + x];
                }
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int r0 = dy * 8 + (t4 >> 1) * 4 + (t4 & 1) * 2;
                    v[r0] = (sm[dy][0] + sm[dy][1]) + sm[dy][2];
                    v[r0 + 1] = (sm[dy][1] - sm[dy][2]) - sm[dy][3];
                }
            }
            // from here on: the direct halo kernel's epilogue for row tile q, column tile j (same pixel per element r)
            const int col = n0 + j * 32 + c31;
            const float bv = col < p.Cout ? p.bias[col] : 0.f;
            double gs = 0.0, gss = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = v[r] + bv;
                if (ra) y = y + rv[r];
                if (rb) y = y + r2[r];
                if (ok_l(j, r)) {
                    stg_u32(p.out + uoff(j, r), 4u * loff, y);
                    if (gnp) {
                        const double d = (double)y;
                        gs = gs + d;
                        gss = __builtin_fma(d, d, gss);
                    }
                }
            }
            if (ra && next) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = rn[r];
            }
            if (gnp) {      // levels 1 and 2 of the moment tree (lane halves, channels of the group), as in conv3x3_halo_kernel
                double a = gs + __shfl_xor(gs, 32, 64), b = gss + __shfl_xor(gss, 32, 64);
                for (int d = 1; d < cg; d <<= 1) {
                    a = a + __shfl_xor(a, d, 64);
                    b = b + __shfl_xor(b, d, 64);
                }
                const int cl = j * 32 + c31;
                if (lane < 32 && (cl & (cg - 1)) == 0) {
                    double *dst = red + ((size_t)q * gpb + cl / cg) * 2;
                    dst[0] = a;
                    dst[1] = b;
                }
            }
        }
        __syncthreads();
        WTT(6)
    }
    WTT_END
    if (gnp && t < gpb) {
        const int g = n0 / cg + t;
        if (g < 32) {
            double S = red[(0 * gpb + t) * 2], SS = red[(0 * gpb + t) * 2 + 1];
#pragma unroll
            for (int qq = 1; qq < 4; ++qq) {           // level 3: ((q0 + q1) + q2) + q3
                S = S + red[((size_t)qq * gpb + t) * 2];
                SS = SS + red[((size_t)qq * gpb + t) * 2 + 1];
            }
            double *dst = p.gn_part + (((size_t)n * p.tilesY * p.tilesX + (size_t)ty * p.tilesX + tx) * 32 + g) * 2;
            dst[0] = S;
            dst[1] = SS;
        }
    }
}

// 3x3 OIHW -> U = G g G^T, K order of a 4x4-tap conv (k = ((ci/32)*16 + 4 i + j)*32 + ci%32), stored as
// out[q = k/32][ntile][g][lane][t]: column o = 32 ntile + lane%32, k = 32 q + 2 (4 g + t) + lane/32   (same size as the
// fragment-major layout of femasr_repack_oihw; a wave's 16-byte-per-lane load of one (q, ntile, g) is contiguous)
__global__ void repack_wino_kernel(const float *__restrict__ in, int O, int I, float *__restrict__ out, size_t total)
{
    const int K = I * 16, NT32 = (O + 31) / 32;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int tq = (int)(idx & 3), lane = (int)((idx >> 2) & 63), g = (int)((idx >> 8) & 3);
        const size_t rest = idx >> 10;
        const int ntile = (int)(rest % NT32), q = (int)(rest / NT32);
        const int k = q * 32 + (4 * g + tq) * 2 + (lane >> 5), o = ntile * 32 + (lane & 31);
        float v = 0.f;
        if (k < K && o < O) {
            const int cl = k % 32, r = k / 32, comp = r & 15, ci = (r >> 4) * 32 + cl;
            const int i = comp >> 2, j = comp & 3;
            const float *gw = in + ((size_t)o * I + ci) * 9;
            float u[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const float g0 = gw[x], g1 = gw[3 + x], g2 = gw[6 + x];
                u[x] = i == 0 ? g0 : (i == 1 ? ((g0 + g1) + g2) * 0.5f : (i == 2 ? ((g0 - g1) + g2) * 0.5f : g2));
            }
            v = j == 0 ? u[0] : (j == 1 ? ((u[0] + u[1]) + u[2]) * 0.5f : (j == 2 ? ((u[0] - u[1]) + u[2]) * 0.5f : u[2]));
        }
        out[idx] = v;
    }
}

struct WVariant {
    const char *name;
    int bn;
    void (*kern)(const ConvParams);
    size_t lds;
    unsigned long long attr_devs;
};
#define FEMASR_WINO(TNW, PRO) { "conv3x3_wino<8x16x" #TNW "*32," #PRO ",waves=8>", 32 * TNW, conv3x3_wino_kernel<TNW, PRO>, wino_lds_bytes<TNW>(), 0ull }
WVariant g_wv[] = {
    FEMASR_WINO(4, FEMASR_PRO_NONE),
    FEMASR_WINO(4, FEMASR_PRO_GN_SILU),
    FEMASR_WINO(2, FEMASR_PRO_NONE),
    FEMASR_WINO(2, FEMASR_PRO_GN_SILU),
};
constexpr int kNumW = sizeof(g_wv) / sizeof(g_wv[0]);

}  // namespace

bool femasr_conv_wino_shape_ok(const femasr_conv_args *a)
{
    return a->ksz == 3 && a->stride == 1 && a->pad == 1 && !a->up2 && a->act == FEMASR_ACT_NONE && (a->Cin % BK) == 0 &&
           (a->Cout % 64) == 0 && (a->prologue == FEMASR_PRO_NONE || a->prologue == FEMASR_PRO_GN_SILU) &&
           (size_t)a->B * a->H * a->W * a->Cin < ((size_t)1 << 31) && (size_t)a->B * a->H * a->W * a->Cout < ((size_t)1 << 31);
}
int femasr_conv_wino_variant_count() { return kNumW; }
const char *femasr_conv_wino_variant_name(int v) { return v >= 0 && v < kNumW ? g_wv[v].name : "?"; }

int femasr_conv_wino_launch(hipStream_t s, const femasr_conv_args *a, int *variant_out, double *flops_out)
{
    FEMASR_REQUIRE(a && a->in && a->w_wino && a->bias && a->out && femasr_conv_wino_shape_ok(a), "conv_wino: bad arguments / shape");
    FEMASR_REQUIRE(a->Ho == a->H && a->Wo == a->W, "conv_wino: Ho/Wo mismatch");
    if (a->prologue == FEMASR_PRO_GN_SILU) FEMASR_REQUIRE(a->pro_a && a->pro_b, "conv_wino: GN prologue needs a,b");
    FEMASR_REQUIRE(!a->gn_part || femasr_gn_fusable(a->Cout), "conv_wino: gn_part needs 32 | Cout and Cout/32 a power of two <= 32");
    ConvParams p{};
    p.in = a->in; p.w_wino = (const float *)a->w_wino; p.bias = a->bias; p.pro_a = a->pro_a; p.pro_b = a->pro_b;
    p.res1 = a->res1; p.res2 = a->res2; p.out = a->out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.ksz = 3; p.stride = 1; p.pad = 1; p.Ho = a->H; p.Wo = a->W;
    p.NT32 = (a->Cout + 31) / 32;
    p.gn_part = a->gn_part;
    const int vi = ((a->Cout % 128) == 0 ? 0 : 2) + (a->prologue == FEMASR_PRO_GN_SILU ? 1 : 0);
    WVariant &v = g_wv[vi];
    p.tilesX = (a->W + 15) / 16;
    p.tilesY = (a->H + 7) / 8;
    p.MB = a->B * p.tilesX * p.tilesY;
    p.NB = a->Cout / v.bn;
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((v.attr_devs >> dev) & 1ull)) {
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
        if (dev >= 0 && dev < 64) v.attr_devs |= 1ull << dev;
    }
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(p.MB * p.NB)), dim3(WI_NT), v.lds, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (variant_out) *variant_out = vi;
    // ALGORITHMIC flops (the definition's 9 taps per output pixel, like every other conv launcher); the kernel issues 16/36 of them
    if (flops_out) *flops_out = 2.0 * (double)a->B * a->H * a->W * 9.0 * (double)a->Cin * (double)a->Cout;
    return FEMASR_OK;
}

extern "C" {

#ifdef FEMASR_WINO_TT
int femasr_debug_wino_time(unsigned long long *buf, int reset)
{
    if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wi_tt), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(buf, HIP_SYMBOL(g_wi_tt), 8 * sizeof(unsigned long long));
}
#endif

size_t femasr_wino_weight_floats(int O, int I) { return (I % 32) == 0 ? femasr_packed_weight_floats(O, I, 4, 4) : 0; }

int femasr_repack_oihw_wino(void *stream, const float *in, int O, int I, float *out)
{
    FEMASR_REQUIRE(in && out && O > 0 && I > 0 && (I % 32) == 0, "repack_wino: needs a 3x3 OIHW weight with I %% 32 == 0");
    const size_t total = femasr_packed_weight_floats(O, I, 4, 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(repack_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, O, I, out, total);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

}  // extern "C"
