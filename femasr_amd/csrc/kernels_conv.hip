// kernels_conv.hip — NHWC implicit-GEMM convolution / linear on the gfx950 fp32 matrix cores.
//
// One kernel family covers every dense contraction of the hot path:
//   nn.Conv2d k1/k3/k4, stride 1/2, zero padding      femasr_arch.py:150,159,173,203,273,298;
//                                                     fema_utils.py:75,78,90; network_swinir.py:465
//   nn.Upsample(scale_factor=2) fused into the loader femasr_arch.py:172,202
//   nn.Linear (ksz=1 on a (1,rows,1,Cin) tensor)      network_swinir.py:19-21,105-112
//   GroupNorm-apply + SiLU on load (PRO_GN)           fema_utils.py:72-79
//   LayerNorm-apply on load (PRO_LN)                  network_swinir.py:243,277
//   bias, GELU(erf), up to two residual adds on store fema_utils.py:82-83; network_swinir.py:276-277,482;
//                                                     femasr_arch.py:361-362
//   VQ distance + per-tile first-min argmin epilogue  femasr_arch.py:35-38,63-66
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = ksz*ksz*Cin in (ky,kx,cin) order.
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32, and per output element ONE k-ascending fmaf
// chain, which is the order the oracle specifies, so results are bit-identical to it.  That
// fixes the schedule: no split-K, one accumulator per output, K walked in ascending order.
//
// Tiling: 256 threads = 4 waves (one per SIMD), block tile BM x BN, BK = 32.
//   A tile (BM x 32) staged global -> registers -> (prologue math) -> LDS [m][33] (pad: conflict-free
//   ds_read_b32 of the MFMA A fragment A[i=lane&31][k=lane>>5] AND conflict-free scattered stores);
//   B tile (32 x BN) = weights [k][n], stored as loaded ([k][n], ds_write_b128), fragment reads of
//   B[k=lane>>5][j=lane&31] hit 32 consecutive banks.
//   Double-buffered LDS, one __syncthreads per K-chunk: loads of chunk c+1 are issued before the
//   16 k-pair MFMA steps of chunk c and written to the other buffer after them.
//   fp32 MFMA is 64 cycles/instruction/SIMD, so LDS and the VALU prologue run in its shadow.
// Grid: 1-D over (m-block, n-block), n fastest, with the bijective XCD remap so that the blocks
//   sharing an A tile / neighbouring halo rows land on the same XCD's L2.
#include "common.h"
#include "detmath.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvParams {
    const float *in, *w, *bias, *pro_a, *pro_b, *pro_c, *res1, *res2;
    float *out;
    const float *vq_zz, *vq_ee;
    float *vq_part;
    int vq_nblk;
    int B, H, W, Cin, Cout, ksz, stride, pad, up2, act, Ho, Wo;
    int M, K, nchunks, taps, MB, NB;
    int tilesX, tilesY;
};

constexpr int BK = 32;
constexpr int ALD = BK + 1;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int BM, int BN, int WM, int WN, int PRO, bool CINVEC, bool VQ, bool WVEC>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 4 : 2) void conv_igemm_kernel(const ConvParams p)
{
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NT = WM * WN * 64;             // 256 (4 waves) or 512 (8 waves, 2 per SIMD inside the block)
    constexpr int RSTEP = NT / 8;                // A rows covered per unit round
    constexpr int AROWS = BM / RSTEP;            // A float4 units per thread
    constexpr int BUNITS = (BK * BN / 4) / NT;   // B float4 units per thread
    static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && BUNITS >= 1 && AROWS >= 1, "tile config");
    static_assert(CINVEC || PRO == FEMASR_PRO_NONE, "generic-Cin path has no prologue");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [2][BM][ALD]
    float *Bs = smem + 2 * BM * ALD;   // [2][BK][BN]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware, bijective block -> tile remap (block b runs on XCD b % 8)
    int L;
    {
        const int nblk = p.MB * p.NB, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int nb = L % p.NB, mb = L / p.NB;
    const int m0 = mb * BM, n0 = nb * BN;

    // ---- per-thread A rows: (mrow + 32 j, k-quad kq)
    const int kq = t & 7, mrow = t >> 3;
    int rn[AROWS], riy[AROWS], rix[AROWS];
    float lmean[AROWS], lrstd[AROWS];
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
            if (PRO == FEMASR_PRO_LN) {
                lmean[j] = p.pro_a[2 * (size_t)r];
                lrstd[j] = p.pro_a[2 * (size_t)r + 1];
            }
        } else {
            rn[j] = 0;
            riy[j] = -(1 << 28);
            rix[j] = -(1 << 28);
            if (PRO == FEMASR_PRO_LN) { lmean[j] = 0.f; lrstd[j] = 0.f; }
        }
    }
    const int Hv = p.up2 ? 2 * p.H : p.H, Wv = p.up2 ? 2 * p.W : p.W;

    float4 ra[AROWS], rga[AROWS], rgb[AROWS], rb[BUNITS];
    float4 lng, lnb;
    unsigned amask = 0, bmask = 0;

    // ---- global -> register staging of K-chunk c
    auto load_chunk = [&](int c) {
        amask = 0;
        bmask = 0;
        if (CINVEC) {
            const int cc = c / p.taps, tap = c - cc * p.taps, c0 = cc * BK + 4 * kq;   // K order: (cin/32, ky, kx, cin%32)
            const int ky = tap / p.ksz, kx = tap - ky * p.ksz;
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                // branch-free: out-of-image taps / tail rows read element 0 and are zeroed at store time,
                // so all loads of a chunk issue back-to-back and stay in flight across the MFMA steps
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
            if (PRO == FEMASR_PRO_LN) {
                lng = ld4(p.pro_b + c0);
                lnb = ld4(p.pro_c + c0);
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
#pragma unroll
        for (int u = 0; u < BUNITS; ++u) {
            const int unit = t + NT * u;
            const int nq = unit % (BN / 4), kr = unit / (BN / 4);
            const int k = c * BK + kr, n = n0 + 4 * nq;
            float4 v;
            if (WVEC) {
                // Cout % 4 == 0: one 16-B load; rows past K / columns past Cout read row 0 and are zeroed
                const bool ok = (k < p.K) & (n < p.Cout);
                v = ld4(p.w + (ok ? ((size_t)k * p.Cout + n) : (size_t)0));
                bmask |= (ok ? 1u : 0u) << u;          // zeroed at store time (keeps the load in flight)
            } else {
                v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) {
                    const float *wp = p.w + (size_t)k * p.Cout + n;
                    if (n + 0 < p.Cout) v.x = wp[0];
                    if (n + 1 < p.Cout) v.y = wp[1];
                    if (n + 2 < p.Cout) v.z = wp[2];
                    if (n + 3 < p.Cout) v.w = wp[3];
                }
            }
            rb[u] = v;
        }
    };

    // ---- registers -> LDS (with the fused normalisation / activation prologue)
    auto store_chunk = [&](int buf) {
        float *Ab = As + buf * BM * ALD;
        float *Bb = Bs + buf * BK * BN;
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            float4 v = ra[j];
            if (PRO == FEMASR_PRO_GN_SILU) {
                v.x = det_silu(__builtin_fmaf(v.x, rga[j].x, rgb[j].x));
                v.y = det_silu(__builtin_fmaf(v.y, rga[j].y, rgb[j].y));
                v.z = det_silu(__builtin_fmaf(v.z, rga[j].z, rgb[j].z));
                v.w = det_silu(__builtin_fmaf(v.w, rga[j].w, rgb[j].w));
            } else if (PRO == FEMASR_PRO_LN) {
                v.x = __builtin_fmaf((v.x - lmean[j]) * lrstd[j], lng.x, lnb.x);
                v.y = __builtin_fmaf((v.y - lmean[j]) * lrstd[j], lng.y, lnb.y);
                v.z = __builtin_fmaf((v.z - lmean[j]) * lrstd[j], lng.z, lnb.z);
                v.w = __builtin_fmaf((v.w - lmean[j]) * lrstd[j], lng.w, lnb.w);
            }
            if (!(amask & (1u << j))) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies AFTER the activation
            float *dst = Ab + (mrow + RSTEP * j) * ALD + 4 * kq;
            dst[0] = v.x;
            dst[1] = v.y;
            dst[2] = v.z;
            dst[3] = v.w;
        }
#pragma unroll
        for (int u = 0; u < BUNITS; ++u) {
            const int unit = t + NT * u;
            const int nq = unit % (BN / 4), kr = unit / (BN / 4);
            float4 v = rb[u];
            if (WVEC && !(bmask & (1u << u))) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(Bb + kr * BN + 4 * nq) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int arow = (wm * TM * 32 + (lane & 31)) * ALD + (lane >> 5);
    const int bcol = (lane >> 5) * BN + wn * TN * 32 + (lane & 31);

    for (int c = 0; c < p.nchunks; ++c) {
        const int buf = c & 1;
        const bool more = (c + 1) < p.nchunks;
        if (more) load_chunk(c + 1);
        const float *Ab = As + buf * BM * ALD + arow;
        const float *Bb = Bs + buf * BK * BN + bcol;
        // fragments of k-pair kk+1 are fetched from LDS before the MFMAs of k-pair kk are issued, so one wave
        // alone keeps its SIMD's matrix pipe fed (the LDS latency hides behind 4 x 64 cycles of MFMA)
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[i * 32 * ALD];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bb[j * 32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = Ab[i * 32 * ALD + 2 * (kk + 1)];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = Bb[2 * (kk + 1) * BN + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of this k-pair's MFMAs
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (!VQ) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
                const float bv = col < p.Cout ? p.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < p.M && col < p.Cout) {
                        float v = acc[i][j][r] + bv;
                        if (p.act == FEMASR_ACT_GELU) v = det_gelu(v);
                        const size_t o = (size_t)row * p.Cout + col;
                        if (p.res1) v = v + p.res1[o];
                        if (p.res2) v = v + p.res2[o];
                        p.out[o] = v;
                    }
                }
            }
    } else {
        // d = (|z|^2 + |e|^2) - 2 z.e ; first-min over this block's BN columns, per row.
        float *red = smem;   // [WN][BM][2]  (A/B buffers are dead after the loop's last barrier)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowl = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int row = m0 + rowl;
                const float zz = row < p.M ? p.vq_zz[row] : 0.f;
                float bd = INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
                    if (col < p.Cout) {
                        const float d = (zz + p.vq_ee[col]) - 2.0f * acc[i][j][r];
                        if (d < bd || (d == bd && col < bi)) { bd = d; bi = col; }
                    }
                }
#pragma unroll
                for (int s = 16; s >= 1; s >>= 1) {
                    const float od = __shfl_xor(bd, s, 64);
                    const int oi = __shfl_xor(bi, s, 64);
                    if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
                }
                if ((lane & 31) == 0) {
                    red[(wn * BM + rowl) * 2] = bd;
                    red[(wn * BM + rowl) * 2 + 1] = __int_as_float(bi);
                }
            }
        __syncthreads();
        if (t < BM && m0 + t < p.M) {   // (BM <= NT)
            float bd = red[t * 2];
            int bi = __float_as_int(red[t * 2 + 1]);
#pragma unroll
            for (int w2 = 1; w2 < WN; ++w2) {
                const float od = red[(w2 * BM + t) * 2];
                const int oi = __float_as_int(red[(w2 * BM + t) * 2 + 1]);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            float *dst = p.vq_part + ((size_t)(m0 + t) * p.vq_nblk + nb) * 2;
            dst[0] = bd;
            dst[1] = __int_as_float(bi);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// 3x3 / stride-1 convolution with HALO RE-USE (the 753 of 964 GFLOP per tile that are 3x3 convs).
// A block owns an 8 x 16 patch of output pixels of ONE image (BM = 128) and BN output channels.  For
// each 32-channel block of the input it stages the (8+2) x (16+2) input halo patch (6 x 10 low-res
// pixels when the nearest-x2 upsample is fused) ONCE — GroupNorm-apply + SiLU evaluated once per
// staged element — and sweeps all 9 taps over it straight from LDS: the MFMA A-fragment of tap
// (ky,kx) is the same LDS image read at a shifted pixel.  Versus the im2col kernel above this cuts
// global loads, prologue VALU work and LDS stores per MFMA by 6.4x (9 x 128 -> 180 pixel-chunks).
// K order = (cin/32, ky, kx, cin%32): exactly the oracle's blocked fmaf chain.
// Pipeline: weights of the next (channel-block, tap) and — at tap 0 — the next channel block's
// patch are loaded to registers before the 16 MFMA k-pair steps and written to the alternate LDS
// buffers after them; one barrier per tap.
// ------------------------------------------------------------------------------------------------
template <int BN, int WM, int WN, int PRO, bool UP2, bool WVEC>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 4 : 2) void conv3x3_halo_kernel(const ConvParams p)
{
    constexpr int BM = 128, TW = 16;
    constexpr int PH = UP2 ? 6 : 10, PW = UP2 ? 10 : 18, PP = PH * PW;
    constexpr int NT = WM * WN * 64;              // 256 (4 waves) or 512 (8 waves: 2 per SIMD inside the block)
    constexpr int PUNITS = (PP * 8 + NT - 1) / NT;
    constexpr int PROWS = NT / 8;                 // patch pixels covered per unit round
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int BUNITS = (BK * BN / 4) / NT;
    static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && BUNITS >= 1, "tile config");
    static_assert(PRO != FEMASR_PRO_LN, "no LayerNorm prologue on 3x3 convs");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PSZ = ((PP * ALD + 3) / 4) * 4;
    float *Ps = smem;               // [2][PP][ALD]
    float *Bs = smem + 2 * PSZ;     // [2][BK][BN]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    int L;
    {
        const int nblk = p.MB * p.NB, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int nb = L % p.NB;
    int tile = L / p.NB;
    const int tx = tile % p.tilesX;
    tile /= p.tilesX;
    const int ty = tile % p.tilesY;
    const int n = tile / p.tilesY;
    const int oy0 = ty * 8, ox0 = tx * TW, n0 = nb * BN;
    const int sy0 = UP2 ? (oy0 >> 1) - 1 : oy0 - 1, sx0 = UP2 ? (ox0 >> 1) - 1 : ox0 - 1;

    // ---- this thread's patch units: (pixel = (t>>3) + 32 i, channel quad kq)
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

    float4 rp[PUNITS], rb[BUNITS], ga, gb;
    unsigned bmask = 0;

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
        if (PRO == FEMASR_PRO_GN_SILU) {
            v.x = det_silu(__builtin_fmaf(v.x, ga.x, gb.x));
            v.y = det_silu(__builtin_fmaf(v.y, ga.y, gb.y));
            v.z = det_silu(__builtin_fmaf(v.z, ga.z, gb.z));
            v.w = det_silu(__builtin_fmaf(v.w, ga.w, gb.w));
        }
        if (!(pmask & (1u << i))) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding AFTER the activation
        float *dst = Pb + pix * ALD + 4 * kq;
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PUNITS; ++i) store_patch_unit(buf, i);
    };
    auto load_w = [&](int q) {      // q = cc * 9 + tap : rows q*32 .. q*32+31 of the repacked weights
        bmask = 0;
#pragma unroll
        for (int u = 0; u < BUNITS; ++u) {
            const int unit = t + NT * u;
            const int nq = unit % (BN / 4), kr = unit / (BN / 4);
            const int k = q * BK + kr, nn = n0 + 4 * nq;
            float4 v;
            if (WVEC) {
                const bool ok = nn < p.Cout;
                v = ld4(p.w + (ok ? ((size_t)k * p.Cout + nn) : (size_t)0));
                bmask |= (ok ? 1u : 0u) << u;
            } else {
                v = make_float4(0.f, 0.f, 0.f, 0.f);
                const float *wp = p.w + (size_t)k * p.Cout + nn;
                if (nn + 0 < p.Cout) v.x = wp[0];
                if (nn + 1 < p.Cout) v.y = wp[1];
                if (nn + 2 < p.Cout) v.z = wp[2];
                if (nn + 3 < p.Cout) v.w = wp[3];
            }
            rb[u] = v;
        }
    };
    auto store_w_unit = [&](int buf, int u) {
        float *Bb = Bs + buf * BK * BN;
        const int unit = t + NT * u;
        const int nq = unit % (BN / 4), kr = unit / (BN / 4);
        float4 v = rb[u];
        if (WVEC && !(bmask & (1u << u))) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(Bb + kr * BN + 4 * nq) = v;
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < BUNITS; ++u) store_w_unit(buf, u);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ncc = p.Cin / BK;
    load_patch(0);
    load_w(0);
    store_patch(0);
    store_w(0);
    __syncthreads();

    // lane's output pixels: m_i = (wm*TM + i)*32 + (lane&31) -> (py, px) = (m >> 4, m & 15)
    int py[TM], px;
    {
        const int m = wm * TM * 32 + (lane & 31);
        px = m & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i) py[i] = (m >> 4) + 2 * i;
    }
    const int bcol = (lane >> 5) * BN + wn * TN * 32 + (lane & 31);

    for (int cc = 0; cc < ncc; ++cc) {
        const float *Pb = Ps + (cc & 1) * PSZ + (lane >> 5);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int q = cc * 9 + tap;
            const bool more_w = (q + 1) < ncc * 9;
            const bool more_p = (tap == 0) && (cc + 1 < ncc);
            const int ky = tap / 3, kx = tap - ky * 3;
            int aidx[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int prow = UP2 ? ((py[i] + ky - 1) >> 1) + 1 : py[i] + ky;
                const int pcol = UP2 ? ((px + kx - 1) >> 1) + 1 : px + kx;
                aidx[i] = (prow * PW + pcol) * ALD;
            }
            const float *Bb = Bs + (q & 1) * BK * BN + bcol;
            // The staging work of the NEXT tap (weights) / channel block (patch) is cut into slices that sit
            // BETWEEN the 16 MFMA k-pair steps of this tap, so VALU / LDS-store / global-load issue overlaps
            // the matrix pipe inside every wave instead of alternating with it in lock-step across the CU.
            constexpr int KW0 = BK / 2 - BUNITS;          // weight-tile stores in the last BUNITS steps
            constexpr int KP0 = KW0 - PUNITS;             // patch units just before them
            static_assert(KP0 >= 2, "not enough k-pair steps to hide the staging slices");
            float af[2][TM], bf[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = Pb[aidx[i]];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[0][j] = Bb[j * 32];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < BK / 2) {     // prefetch the next k-pair's fragments before issuing this one's MFMAs
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[nxt][i] = Pb[aidx[i] + 2 * (kk + 1)];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[nxt][j] = Bb[2 * (kk + 1) * BN + j * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
                if (kk == 0) {
                    if (more_w) load_w(q + 1);
                    if (more_p) load_patch(cc + 1);
                }
                if (kk >= KP0 && kk < KW0) {
                    if (more_p) store_patch_unit((cc + 1) & 1, kk - KP0);
                }
                if (kk >= KW0) {
                    if (more_w) store_w_unit((q + 1) & 1, kk - KW0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    }

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
                if (oy < p.Ho && ox < p.Wo && col < p.Cout) {
                    float v = acc[i][j][r] + bv;
                    const size_t o = (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout + col;
                    if (p.res1) v = v + p.res1[o];
                    if (p.res2) v = v + p.res2[o];
                    p.out[o] = v;
                }
            }
        }
}

template <int BN, bool UP2>
constexpr size_t halo_lds_bytes()
{
    return (size_t)(2 * ((((UP2 ? 60 : 180) * ALD + 3) / 4) * 4) + 2 * BK * BN) * sizeof(float);
}

template <int BM, int BN>
constexpr size_t conv_lds_bytes() { return (size_t)(2 * BM * ALD + 2 * BK * BN) * sizeof(float); }

struct Variant {
    const char *name;
    int bm, bn;
    void (*kern)(const ConvParams);
    size_t lds;
    bool attr_set;
    int threads;
};

#define FEMASR_VARIANT(BM, BN, WM, WN, PRO, VEC, VQ, WVEC)                                             \
    { "conv_igemm<" #BM "x" #BN "," #PRO ",cinvec=" #VEC ",vq=" #VQ ",wvec=" #WVEC ",waves=" #WM "x" #WN ">", BM, BN,        \
      conv_igemm_kernel<BM, BN, WM, WN, PRO, VEC, VQ, WVEC>, conv_lds_bytes<BM, BN>(), false, WM * WN * 64 }

#define FEMASR_HALO(BN, WM, WN, PRO, UP2, WVEC)                                                        \
    { "conv3x3_halo<8x16x" #BN "," #PRO ",up2=" #UP2 ",wvec=" #WVEC ",waves=" #WM "x" #WN ">", 128, BN,                      \
      conv3x3_halo_kernel<BN, WM, WN, PRO, UP2, WVEC>, halo_lds_bytes<BN, UP2>(), false, WM * WN * 64 }

Variant g_variants[] = {
    FEMASR_VARIANT(128, 128, 4, 2, FEMASR_PRO_NONE, true, false, true),     // 0 (8 waves)
    FEMASR_VARIANT(128, 128, 4, 2, FEMASR_PRO_GN_SILU, true, false, true),  // 1
    FEMASR_VARIANT(128, 128, 4, 2, FEMASR_PRO_LN, true, false, true),       // 2
    FEMASR_VARIANT(128, 64, 4, 1, FEMASR_PRO_NONE, true, false, true),      // 3
    FEMASR_VARIANT(128, 64, 4, 1, FEMASR_PRO_GN_SILU, true, false, true),   // 4
    FEMASR_VARIANT(128, 64, 4, 1, FEMASR_PRO_LN, true, false, true),        // 5
    FEMASR_VARIANT(128, 32, 4, 1, FEMASR_PRO_NONE, true, false, false),     // 6  any Cout (out_conv: Cout = 3)
    FEMASR_VARIANT(128, 32, 4, 1, FEMASR_PRO_GN_SILU, true, false, false),  // 7
    FEMASR_VARIANT(128, 32, 4, 1, FEMASR_PRO_LN, true, false, false),       // 8
    FEMASR_VARIANT(128, 128, 2, 2, FEMASR_PRO_NONE, false, false, true),    // 9  generic Cin (in_conv)
    FEMASR_VARIANT(128, 64, 4, 1, FEMASR_PRO_NONE, false, false, true),     // 10
    FEMASR_VARIANT(128, 32, 4, 1, FEMASR_PRO_NONE, false, false, false),    // 11
    FEMASR_VARIANT(128, 128, 2, 2, FEMASR_PRO_NONE, true, true, true),      // 12 VQ distance + argmin
    FEMASR_HALO(128, 4, 2, FEMASR_PRO_NONE, false, true),                   // 13 3x3 s1 halo kernels
    FEMASR_HALO(128, 4, 2, FEMASR_PRO_GN_SILU, false, true),                // 14 (8 waves)
    FEMASR_HALO(128, 4, 2, FEMASR_PRO_NONE, true, true),                    // 15 fused nearest-x2
    FEMASR_HALO(64, 4, 2, FEMASR_PRO_NONE, false, true),                    // 16
    FEMASR_HALO(64, 4, 2, FEMASR_PRO_GN_SILU, false, true),                 // 17 (8 waves)
    FEMASR_HALO(64, 4, 2, FEMASR_PRO_NONE, true, true),                     // 18
    FEMASR_HALO(32, 4, 1, FEMASR_PRO_NONE, false, false),                   // 19 any Cout (out_conv)
    FEMASR_HALO(32, 4, 1, FEMASR_PRO_GN_SILU, false, false),                // 20
    FEMASR_HALO(32, 4, 1, FEMASR_PRO_NONE, true, false),                    // 21
};
constexpr int kNumVariants = sizeof(g_variants) / sizeof(g_variants[0]);

bool use_halo(const femasr_conv_args *a, bool vq)
{
    return !vq && a->ksz == 3 && a->stride == 1 && a->pad == 1 && (a->Cin % BK) == 0 && a->prologue != FEMASR_PRO_LN &&
           a->act == FEMASR_ACT_NONE && !(a->up2 && a->prologue != FEMASR_PRO_NONE) &&
           (size_t)a->B * a->H * a->W * a->Cin < ((size_t)1 << 31);
}

int pick_variant(const femasr_conv_args *a, bool vq)
{
    const bool vec = (a->Cin % BK) == 0;
    if (vq) return 12;
    // BN by Cout; the 16-byte weight loads need Cout % 4 == 0, anything else goes to the BN=32 scalar-load variants
    const int cls = (a->Cout & 3) ? 2 : (a->Cout > 64 ? 0 : (a->Cout > 32 ? 1 : 2));
    if (use_halo(a, vq)) return 13 + cls * 3 + (a->up2 ? 2 : a->prologue);
    if (!vec) return 9 + cls;
    return cls * 3 + a->prologue;
}

}  // namespace

int femasr_conv_variant_count() { return kNumVariants; }
const char *femasr_conv_variant_name(int v) { return (v >= 0 && v < kNumVariants) ? g_variants[v].name : "?"; }

int femasr_conv2d_launch(hipStream_t s, const femasr_conv_args *a, const conv_vq_epilogue *vq,
                         int *variant_out, double *flops_out)
{
    FEMASR_REQUIRE(a && a->in && a->w && a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0,
                   "conv2d: null pointer or empty shape");
    FEMASR_REQUIRE(vq || (a->bias && a->out), "conv2d: bias/out must be set");
    FEMASR_REQUIRE(a->ksz >= 1 && a->ksz <= 7 && (a->stride == 1 || a->stride == 2) && a->pad >= 0,
                   "conv2d: unsupported ksz=%d stride=%d pad=%d", a->ksz, a->stride, a->pad);
    FEMASR_REQUIRE(a->prologue >= 0 && a->prologue <= 2, "conv2d: bad prologue %d", a->prologue);
    const int Hv = a->up2 ? 2 * a->H : a->H, Wv = a->up2 ? 2 * a->W : a->W;
    const int Ho = (Hv + 2 * a->pad - a->ksz) / a->stride + 1, Wo = (Wv + 2 * a->pad - a->ksz) / a->stride + 1;
    FEMASR_REQUIRE(Ho == a->Ho && Wo == a->Wo, "conv2d: Ho/Wo mismatch (%d,%d) vs expected (%d,%d)", a->Ho, a->Wo, Ho, Wo);
    const bool vec = (a->Cin % BK) == 0;
    FEMASR_REQUIRE(vec || a->prologue == FEMASR_PRO_NONE, "conv2d: prologue needs Cin %% 32 == 0 (Cin=%d)", a->Cin);
    if (a->prologue == FEMASR_PRO_GN_SILU) FEMASR_REQUIRE(a->pro_a && a->pro_b, "conv2d: GN prologue needs a,b");
    if (a->prologue == FEMASR_PRO_LN) FEMASR_REQUIRE(a->pro_a && a->pro_b && a->pro_c, "conv2d: LN prologue needs stats,gamma,beta");
    const long long M = (long long)a->B * Ho * Wo;
    FEMASR_REQUIRE(M < (1ll << 31) - 256, "conv2d: too many output pixels");

    ConvParams p{};
    p.in = a->in; p.w = a->w; p.bias = a->bias; p.pro_a = a->pro_a; p.pro_b = a->pro_b; p.pro_c = a->pro_c;
    p.res1 = a->res1; p.res2 = a->res2; p.out = a->out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.ksz = a->ksz; p.stride = a->stride;
    p.pad = a->pad; p.up2 = a->up2; p.act = a->act; p.Ho = Ho; p.Wo = Wo;
    p.M = (int)M; p.K = a->ksz * a->ksz * a->Cin; p.nchunks = (p.K + BK - 1) / BK; p.taps = a->ksz * a->ksz;
    const int vi = pick_variant(a, vq != nullptr);
    Variant &v = g_variants[vi];
    p.MB = (p.M + v.bm - 1) / v.bm;
    p.NB = (p.Cout + v.bn - 1) / v.bn;
    if (vi >= 13) {   // halo kernels: 2-D tiles of 8 x 16 output pixels per image
        p.tilesX = (Wo + 15) / 16;
        p.tilesY = (Ho + 7) / 8;
        p.MB = a->B * p.tilesX * p.tilesY;
    }
    if (vq) {
        FEMASR_REQUIRE(vq->zz && vq->ee && vq->part && vq->nblk == p.NB && (a->Cout % 32) == 0, "vq epilogue: bad args");
        p.vq_zz = vq->zz; p.vq_ee = vq->ee; p.vq_part = vq->part; p.vq_nblk = vq->nblk;
    }
    if (!v.attr_set) {
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
        v.attr_set = true;
    }
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(p.MB * p.NB)), dim3((unsigned)v.threads), v.lds, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (variant_out) *variant_out = vi;
    if (flops_out) *flops_out = 2.0 * (double)M * (double)a->Cout * (double)p.K;
    return FEMASR_OK;
}
