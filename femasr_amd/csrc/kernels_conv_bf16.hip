// kernels_conv_bf16.hip — 3x3 halo convolution on the bf16 matrix cores with a 3-term split ("bf16x3").
//
// Where it is used: ONLY behind the VQ codebook lookup (after_quant conv, the three DecoderBlocks, out_conv;
// femasr_arch.py:195-211,267-273,352,366-369), and only when the caller opts in (femasr_set_decoder_math(1)).
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
// pre-split and stored fragment-major by femasr_repack_oihw_bf16x3 and read straight into MFMA operands.
//   LDS patch image: ushort [2 buffers][hi|lo][pixel][40]  (80-byte pixel pitch -> the ds_read_b128 of the A fragment
//   A[i=lane&31][k=8*(lane>>5)..+7] is conflict-free across each 16-lane group)
// Blocks: 256 threads = 4 waves, 64x64 (BN=128) / 32x64 (BN=64) / 32x32 (BN=32) outputs per wave.
#include "conv_common.h"
#include "detmath.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 4 : 2) void conv3x3_halo_bf16x3_kernel(const ConvParams p, const uint4 *__restrict__ wsplit,
                                                                                          double *__restrict__ stats_part)
{
    constexpr int BM = 128, TW = 16, NT = WM * WN * 64;
    constexpr int PH = UP2 ? 6 : 10, PW = UP2 ? 10 : 18, PP = PH * PW;
    constexpr int PUNITS = (PP * 8 + NT - 1) / NT, PROWS = NT / 8;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1, "tile config");
    static_assert(PRO != FEMASR_PRO_LN, "no LayerNorm prologue on 3x3 convs");
    static_assert(1 + PUNITS <= 9, "patch slices are spread over taps 1..");

    extern __shared__ __attribute__((aligned(16))) unsigned short smem_u16[];
    constexpr int HALF = PP * PPITCH;            // ushorts per (buffer, hi|lo) image
    unsigned short *Ps = smem_u16;               // [2][2][PP][PPITCH]

    const int t = threadIdx.x, lane = t & 63;
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
        const int pix = (t >> 3) + PROWS * i;
        if (PP % PROWS != 0 && i == PUNITS - 1 && pix >= PP) return;
        float4 v = rp[i];
        if (PRO == FEMASR_PRO_GN_SILU) {
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

    // accumulators start from bias + residuals (this path is tolerance-based, so the order of the final additions is
    // free): the residual loads overlap the first patch load instead of serialising in the epilogue
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            const float bv = col < p.Cout ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
                float v = bv;
                if (oy < p.Ho && ox < p.Wo && col < p.Cout) {
                    const size_t o = (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout + col;
                    if (p.res1) v = v + p.res1[o];
                    if (p.res2) v = v + p.res2[o];
                }
                acc[i][j][r] = v;
            }
        }

    // split weights, fragment-major: [q][ntile][lane][kstep(2)][hi8|lo8] bf16 = 4 x uint4 per (q, ntile, lane)
    const uint4 *wl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wl[j] = wsplit + ((((size_t)(n0 >> 5) + wn * TN + j) * 64 + lane) << 2);
    const size_t wstride = (size_t)p.NT32 << 8;     // uint4 per K chunk

    const int ncc = p.Cin / BK;
    load_patch(0);
    uint4 bc[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) bc[j][e] = wl[j][e];
#pragma unroll
    for (int i = 0; i < PUNITS; ++i) store_patch_unit(0, i);
    __syncthreads();

    int py[TM], px;
    {
        const int m = wm * TM * 32 + (lane & 31);
        px = m & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i) py[i] = (m >> 4) + 2 * i;
    }
    const int koff = 8 * (lane >> 5);                // this lane's k sub-block inside a 16-deep k-step

    for (int cc = 0; cc < ncc; ++cc) {
        const unsigned short *Pb = Ps + ((cc & 1) * 2) * HALF + koff;
        const bool more_p = cc + 1 < ncc;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int q = cc * 9 + tap;
            const bool more_w = (q + 1) < ncc * 9;
            const int ky = tap / 3, kx = tap - ky * 3;
            if (more_p && tap == 0) load_patch(cc + 1);
            int aidx[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int prow = UP2 ? ((py[i] + ky - 1) >> 1) + 1 : py[i] + ky;
                const int pcol = UP2 ? ((px + kx - 1) >> 1) + 1 : px + kx;
                aidx[i] = (prow * PW + pcol) * PPITCH;
            }
            // A fragments are fetched one k-step ahead (s=1 while s=0 multiplies; the next tap's s=0 is fetched by the
            // next iteration's prologue read below), pinned with sched_barrier so the LDS latency hides behind MFMAs
            uint4 a_hi[2][TM], a_lo[2][TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a_hi[0][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i]);
                a_lo[0][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + HALF);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        a_hi[1][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + 16);
                        a_lo[1][i] = *reinterpret_cast<const uint4 *>(Pb + aidx[i] + 16 + HALF);
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
                // this k-step's weight registers are free now: refill them with the NEXT tap's fragments (half a tap of
                // MFMAs ahead of their use), no register copies
                if (more_w) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bc[j][2 * s] = (wl[j] + (size_t)(q + 1) * wstride)[2 * s];
                        bc[j][2 * s + 1] = (wl[j] + (size_t)(q + 1) * wstride)[2 * s + 1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // next channel block's patch: loaded at tap 0, one unit normalised / split / stored per tap over the LAST
            // PUNITS taps (bf16 taps are ~5x shorter than fp32 ones: the HBM latency needs the distance)
            if (more_p && tap >= 9 - PUNITS) {
#pragma unroll
                for (int i = 0; i < PUNITS; ++i)
                    if (tap == i + 9 - PUNITS) store_patch_unit((cc + 1) & 1, i);
            }
        }
        __syncthreads();
    }

    float colsum[TM][TN], colsq[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            float ps = 0.f, pss = 0.f;      // this lane's share of the GroupNorm moments of the OUTPUT (column `col`)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
                if (oy < p.Ho && ox < p.Wo && col < p.Cout) {
                    const float v = acc[i][j][r];
                    p.out[(((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout + col] = v;
                    ps += v;
                    pss = __builtin_fmaf(v, v, pss);
                }
            }
            if (stats_part) { colsum[i][j] = ps; colsq[i][j] = pss; }
        }

    // Optional fused GroupNorm moments of the output (consumed by the NEXT conv's GN prologue): per (tile, group)
    // partial sums, reduced lane -> group (xor shuffles over the cg lanes of a group, then the two row halves) -> waves
    // (LDS) and written as doubles to stats_part[((n*tiles + tile)*32 + g)*2]; a fixed order, so runs are reproducible.
    if (stats_part) {
        const int cg = p.Cout >> 5;                       // channels per group (32 groups): 8 / 4 / 2
        double *red = reinterpret_cast<double *>(smem_u16);   // [WM][BN/2][2] (patch buffers are dead after the last barrier)
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
                red[(wm * (BN / 2) + gl) * 2] = s_;
                red[(wm * (BN / 2) + gl) * 2 + 1] = q_;
            }
        }
        __syncthreads();
        const int ngl = BN / cg;                                         // groups covered by this block
        if (t < ngl && n0 + t * cg < p.Cout) {
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) {
                S += red[(w2 * (BN / 2) + t) * 2];
                Q += red[(w2 * (BN / 2) + t) * 2 + 1];
            }
            const int g = n0 / cg + t;
            const size_t tile_id = (size_t)n * p.tilesX * p.tilesY + (size_t)ty * p.tilesX + tx;
            stats_part[(tile_id * 32 + g) * 2] = S;
            stats_part[(tile_id * 32 + g) * 2 + 1] = Q;
        }
    }
}

// OIHW fp32 -> split bf16 fragment-major: out ushort index = (((((q*NT32 + ntile)*64 + lane)*2 + s)*2 + h)*8 + e)
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
        const int e = (int)(i & 7), h = (int)((i >> 3) & 1), s = (int)((i >> 4) & 1), lane = (int)((i >> 5) & 63);
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

template <bool UP2>
constexpr size_t bf16_lds_bytes() { return (size_t)2 * 2 * (UP2 ? 60 : 180) * PPITCH * sizeof(unsigned short); }

struct Variant16 {
    const char *name;
    int bn, threads;
    void (*kern)(const ConvParams, const uint4 *, double *);
    size_t lds;
    bool attr_set;
};
#define FEMASR_H16(BN, WM, WN, PRO, UP2)                                                        \
    { "conv3x3_halo_bf16x3<8x16x" #BN "," #PRO ",up2=" #UP2 ",waves=" #WM "x" #WN ">", BN, WM * WN * 64,   \
      conv3x3_halo_bf16x3_kernel<BN, WM, WN, PRO, UP2>, bf16_lds_bytes<UP2>(), false }

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
};
constexpr int kNum16 = sizeof(g_v16) / sizeof(g_v16[0]);

}  // namespace

int femasr_conv_bf16x3_variant_count() { return kNum16; }
const char *femasr_conv_bf16x3_variant_name(int v) { return (v >= 0 && v < kNum16) ? g_v16[v].name : "?"; }

bool femasr_conv_bf16x3_eligible(const femasr_conv_args *a)
{
    return a->w_bf16x3 && a->ksz == 3 && a->stride == 1 && a->pad == 1 && (a->Cin % BK) == 0 &&
           a->prologue != FEMASR_PRO_LN && a->act == FEMASR_ACT_NONE && !(a->up2 && a->prologue != FEMASR_PRO_NONE) &&
           (size_t)a->B * a->H * a->W * a->Cin < ((size_t)1 << 31);
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
    const int cls = a->Cout > 64 ? 0 : (a->Cout > 32 ? 1 : 2);
    const int vi = cls * 3 + (a->up2 ? 2 : a->prologue);
    Variant16 &v = g_v16[vi];
    p.tilesX = (p.Wo + 15) / 16;
    p.tilesY = (p.Ho + 7) / 8;
    p.MB = a->B * p.tilesX * p.tilesY;
    p.NB = (a->Cout + v.bn - 1) / v.bn;
    if (!v.attr_set) {
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
        v.attr_set = true;
    }
    FEMASR_REQUIRE(!a->gn_part || (a->Cout % 32 == 0 && (a->Cout / 32) <= 8 && v.bn % (a->Cout / 32) == 0),
                   "conv bf16x3: fused GN moments need Cout %% 32 == 0 and <= 8 channels per group");
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(p.MB * p.NB)), dim3((unsigned)v.threads), v.lds, s, p, (const uint4 *)a->w_bf16x3,
                       (double *)a->gn_part);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (variant_out) *variant_out = vi;
    if (flops_out) *flops_out = 2.0 * (double)a->B * p.Ho * p.Wo * (double)a->Cout * 9.0 * a->Cin;
    return FEMASR_OK;
}

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
