// kernels_mlp.hip — the Swin MLP (network_swinir.py:14-30,276-277: x + fc2(gelu(fc1(norm2(x))))) as ONE kernel:
//     out[M][256] = res + ( gelu( X[M][256] . W1 + b1 )[M][1024] . W2 + b2 )
// The 1024-wide hidden activation never leaves the CU (round 3 ran fc1 and fc2 as two GEMM launches: 340 MB written and read
// back per block at B = 16).  Every output is the SAME fp32 fmaf chain as in the two-launch form (kernels_gemm.hip), so the
// result is bit-identical to it and to the oracle:
//   * fc1 is computed TRANSPOSED, H^T = W1^T X^T: the packed weight fragment is the MFMA A operand, the activation fragment the
//     B operand (the same registers the GEMM feeds the other way round; products commute, the k order inside the instruction and
//     across the K loop is unchanged).  A lane then holds H[token = lane % 32][hidden = (r & 3) + 8 (r >> 2) + 4 (lane / 32)] in
//     accumulator register r - exactly the A fragment of the NEXT product: register r = 4 j + e supplies k = 8 j + e (lanes 0-31)
//     and 8 j + 4 + e (lanes 32-63), the ORC_KPERM order of the 1x1 layers.  bias + exact-erf GELU run on the accumulators.
//   * fc2 accumulates over the hidden dimension in ascending 32-deep tiles (the GEMM's chunk order), 128 hidden units (one GROUP)
//     at a time: per group 8 K chunks of fc1, then 4 hidden tiles of fc2; Y stays in the accumulators across all 8 groups.
// Block = 128 tokens, 8 waves (2 per SIMD): wave (wm = token tile 0..3, wh = half 0..1) computes fc1 for hidden tiles 2 wh, 2 wh + 1
// of the group and fc2 for output columns 128 wh .. + 127 of its 32 tokens.  The two waves of a token tile swap their GELU'd hidden
// tiles through LDS (a lane reads back, as one 16-byte fragment, what the partner's same lane wrote); its own tiles come from registers.
// Operands reach LDS by LDS-DMA in a flat sequence of 96 32-KB chunks (per group: 8 x [X chunk | W1 chunk], 4 x W2 tile), two stages,
// the counted-wait / two-barrier protocol of gemm_dma<.., ST = 2>.  LDS: 2 x 32 KB stages + 64 KB exchange = 128 KB, one block per CU.
//
// MEASURED (round 4, profiles/r04_mlp_fused.txt): bit-identical at the first run - and 18 % SLOWER than the two launches it replaces
// (M = 82944: 0.944 vs 0.802 ms, 92 vs 108 TFLOP/s; M = 41472: 0.60 vs 0.43 ms).  One 128-KB block per CU means all 8 waves meet at two
// barriers per 32-MFMA chunk with no second block to fill the gaps, the GELU arithmetic sits on the critical path of every wave at the
// same time, and 648 blocks on 256 CUs run 2.53 -> 3 rounds for the WHOLE MLP where fc1 alone (5184 tiles) packs at 99 %.  The hidden
// tensor's HBM round trip it saves was already hidden under the GEMMs' MFMAs.  Kept as an opt-in (FEMASR_MLP=fused) and as the
// parity-tested proof that the transposed-accumulator hand-over works; the forward's default stays the two GEMM launches.
#include "conv_common.h"
#include "detmath.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct MlpParams {
    const float *X, *W1, *b1, *W2, *b2, *res;
    float *out;
    int M;
};

constexpr int ML_C = 256, ML_HID = 1024, ML_BM = 128;
constexpr int ML_STAGE = 8192;                 // floats per stage (32 KB)
constexpr int ML_HBUF = 16384;                 // floats of the hidden-tile exchange (64 KB): [token tile 4][owner half 2][tile 2][j 4][lane 64][4]
constexpr int ML_LDS_BYTES = (2 * ML_STAGE + ML_HBUF) * 4;
constexpr int ML_NCHUNK = 8 * 12;              // 8 groups x (8 fc1 chunks + 4 fc2 tiles)

typedef __attribute__((address_space(3))) void *lds_vptr;
typedef __attribute__((address_space(1))) const void *glb_cvptr;
__device__ __forceinline__ void dma16(const float *gsrc, float *lds_dst_uniform)
{
    __builtin_amdgcn_global_load_lds((glb_cvptr)gsrc, (lds_vptr)lds_dst_uniform, 16, 0, 0);      // 64 lanes x 16 bytes -> 1 KiB at the uniform destination
}

template <int NRES>
__global__ __launch_bounds__(512, 2) void mlp_fused_kernel(const MlpParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Hb = smem + 2 * ML_STAGE;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 3, wh = wave >> 2;
    const int hh = lane >> 5, c31 = lane & 31;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * ML_BM;

    // ---- DMA sources.  X chunk [128 rows][32 k]: 16 pieces of 8 rows, granule g of row r at r*8 + (g ^ ((r >> 1) & 7)) (the A-tile
    // layout of kernels_gemm.hip); this wave copies pieces 2 wave, 2 wave + 1.  W1 chunk of a group: 16 KiB linear, pieces 2 wave ..
    // W2 tile: 32 KiB linear, pieces 4 wave ..
    const float *srcX[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (2 * wave + i) + (lane >> 3);
        const int g = (lane & 7) ^ ((row >> 1) & 7);
        int grow = m0 + row;
        grow = grow < p.M ? grow : p.M - 1;                 // tail rows: clamp (computed, never stored)
        srcX[i] = p.X + (size_t)grow * ML_C + 4 * g;
    }
    const float *srcW1 = p.W1 + (size_t)(2 * wave) * 256 + lane * 4;       // + ((c * 32 + 4 g) * 4) * 256 per (chunk c, group g)
    const float *srcW2 = p.W2 + (size_t)(4 * wave) * 256 + lane * 4;       // + (T * 32) * 256 per hidden tile T
    auto issue = [&](int sb, int q) {
        float *S = smem + sb * ML_STAGE;
        const int g = q / 12, r = q - 12 * g;
        if (r < 8) {
            dma16(srcX[0] + r * 32, S + (2 * wave) * 256);
            dma16(srcX[1] + r * 32, S + (2 * wave + 1) * 256);
            const float *w = srcW1 + (size_t)((r * 32 + 4 * g) * 4) * 256;
            dma16(w, S + 4096 + (2 * wave) * 256);
            dma16(w + 256, S + 4096 + (2 * wave + 1) * 256);
        } else {
            const float *w = srcW2 + (size_t)((4 * g + (r - 8)) * 32) * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(w + i * 256, S + (4 * wave + i) * 256);
        }
    };

    // ---- fragment addresses (floats inside a stage)
    const int xq = hh ^ ((c31 >> 1) & 7);
    int xoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xoff[j] = ((wm * 32 + c31) * 8 + ((2 * j) ^ xq)) * 4;
    const int w1off = 4096 + (wh * 2 * 4) * 256 + lane * 4;        // + (ht * 4 + j) * 256
    const int w2off = (wh * 4 * 4) * 256 + lane * 4;               // + (nt * 4 + j) * 256
    const int hmine = ((wm * 2 + wh) * 2 * 4) * 256 + lane * 4;    // + (ht * 4 + j) * 256: this wave's tiles in the exchange
    const int hpart = ((wm * 2 + (wh ^ 1)) * 2 * 4) * 256 + lane * 4;

    f32x16 accY[4], accH[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accY[i][r] = 0.f;

    issue(0, 0);
    issue(1, 1);

    for (int q = 0; q < ML_NCHUNK; ++q) {
        const int g = q / 12, r = q - 12 * g;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // this wave's pieces of chunk q have landed (those of q + 1 may still fly)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // ... and its exchange writes, if any
        __builtin_amdgcn_s_barrier();                             // ... and every other wave's
        asm volatile("" ::: "memory");
        const float *S = smem + (q & 1) * ML_STAGE;
        if (r < 8) {
            // fc1 chunk: H^T[hidden tiles 2 wh, 2 wh + 1][tokens of wm] += W1^T X^T over 32 k
            if (r == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int k = 0; k < 16; ++k) accH[i][k] = 0.f;
            }
            f32x4_t xf[2], wf[2][2];
            xf[0] = *reinterpret_cast<const f32x4_t *>(S + xoff[0]);
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) wf[0][ht] = *reinterpret_cast<const f32x4_t *>(S + w1off + (ht * 4) * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cur = j & 1, nxt = cur ^ 1;
                if (j + 1 < 4) {
                    xf[nxt] = *reinterpret_cast<const f32x4_t *>(S + xoff[j + 1 < 4 ? j + 1 : 0]);
#pragma unroll
                    for (int ht = 0; ht < 2; ++ht) wf[nxt][ht] = *reinterpret_cast<const f32x4_t *>(S + w1off + (ht * 4 + j + 1) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ht = 0; ht < 2; ++ht)
                        accH[ht] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cur][ht][e], xf[cur][e], accH[ht], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (r == 7) {
                // bias + exact-erf GELU on the accumulators (register k = 4 j + e <-> hidden 8 j + 4 hh + e of the tile), then this
                // wave's two tiles into the exchange as the 16-byte fragments its partner will read
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    const float *bp = p.b1 + g * 128 + (wh * 2 + ht) * 32 + 4 * hh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4_t b4 = *reinterpret_cast<const f32x4_t *>(bp + 8 * j);
                        const det_f32x2 g0 = det_gelu2(det_f32x2{accH[ht][4 * j] + b4[0], accH[ht][4 * j + 1] + b4[1]});
                        const det_f32x2 g1 = det_gelu2(det_f32x2{accH[ht][4 * j + 2] + b4[2], accH[ht][4 * j + 3] + b4[3]});
                        accH[ht][4 * j] = g0[0]; accH[ht][4 * j + 1] = g0[1]; accH[ht][4 * j + 2] = g1[0]; accH[ht][4 * j + 3] = g1[1];
                        *reinterpret_cast<f32x4_t *>(Hb + hmine + (ht * 4 + j) * 256) = f32x4_t{g0[0], g0[1], g1[0], g1[1]};
                    }
                }
            }
        } else {
            // fc2 tile T = 4 g + (r - 8): Y[tokens of wm][columns 128 wh ..] += H[:, tile] W2[tile, :], k ascending inside the tile.
            // The A fragments of the wave's OWN tiles are its accumulator registers as they stand (compile-time register indices: the
            // four (owner, tile) cases are separate code); the partner's come from the exchange.
            const int tt = r - 8;
            auto fc2_tile = [&](auto own_c, auto ht_c) {
                constexpr bool OWN = decltype(own_c)::value;
                constexpr int HT = decltype(ht_c)::value;
                f32x4_t af[4];
                if (!OWN) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f32x4_t *>(Hb + hpart + (HT * 4 + j) * 256);
                }
                f32x4_t bf[2][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bf[0][nt] = *reinterpret_cast<const f32x4_t *>(S + w2off + (nt * 4) * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cur = j & 1, nxt = cur ^ 1;
                    if (j + 1 < 4) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) bf[nxt][nt] = *reinterpret_cast<const f32x4_t *>(S + w2off + (nt * 4 + j + 1) * 256);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            accY[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(OWN ? accH[HT][4 * j + e] : af[j][e], bf[cur][nt][e], accY[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            if ((tt >> 1) == wh) { if (tt & 1) fc2_tile(std::true_type{}, I1{}); else fc2_tile(std::true_type{}, I0{}); }
            else { if (tt & 1) fc2_tile(std::false_type{}, I1{}); else fc2_tile(std::false_type{}, I0{}); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's reads of stage q & 1 have returned ...
        __builtin_amdgcn_s_barrier();                             // ... and every other wave's: the stage may be overwritten
        asm volatile("" ::: "memory");
        issue(q & 1, q + 2 < ML_NCHUNK ? q + 2 : ML_NCHUNK - 1);  // unconditional (the wait count stays constant; the surplus lands in a dead stage)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no DMA may still be writing LDS when the epilogue re-uses it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- epilogue: out = (acc + b2) + res, each 32 x 32 tile transposed through a per-wave LDS scratch so that lane l owns columns
    // 4 (l & 7) .. + 3 of tile rows (l >> 3) + 8 k: 16-byte residual loads and stores (the epilogue of kernels_gemm.hip)
    float *T = smem + wave * TSCRATCH;
    const int trow = lane >> 3, tq = lane & 7;
    const int rbase = m0 + wm * 32;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = wh * 128 + nt * 32 + 4 * tq;
        f32x4_t r1[4];
        if (NRES >= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = rbase + trow + 8 * k;
                r1[k] = *reinterpret_cast<const f32x4_t *>(p.res + (row < p.M ? (size_t)row * ML_C + col : 0));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hh) * TPITCH + c31] = accY[nt][r];
        const f32x4_t b4 = *reinterpret_cast<const f32x4_t *>(p.b2 + col);
        // (same wave wrote and reads the scratch: LDS ops of one wave complete in order)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4_t a4 = *reinterpret_cast<const f32x4_t *>(T + (trow + 8 * k) * TPITCH + 4 * tq);
            float v[4] = {a4[0] + b4[0], a4[1] + b4[1], a4[2] + b4[2], a4[3] + b4[3]};
            if (NRES >= 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + r1[k][e];
            }
            const int row = rbase + trow + 8 * k;
            if (row < p.M) *reinterpret_cast<f32x4_t *>(p.out + (size_t)row * ML_C + col) = f32x4_t{v[0], v[1], v[2], v[3]};
        }
    }
}

unsigned long long g_mlp_attr_devs[2] = {0ull, 0ull};

}  // namespace

const char *femasr_mlp_fused_variant_name() { return "mlp_fused<128 tokens,fc1+gelu+fc2,waves=8>"; }

bool femasr_mlp_fused_shape_ok(int C, int hidden) { return C == ML_C && hidden == ML_HID; }

// x, res, out: (M, 256) row-major; w1p / w2p: fc1.weight (1024, 256) / fc2.weight (256, 1024) in the packed GEMM layout (femasr_repack_k1)
int femasr_mlp_fused_launch(hipStream_t s, const float *x, long long M, const float *w1p, const float *b1, const float *w2p, const float *b2,
                            const float *res, float *out, double *flops_out)
{
    FEMASR_REQUIRE(x && w1p && b1 && w2p && b2 && out && M > 0 && M < (1ll << 31) - 256, "mlp_fused: bad arguments");
    MlpParams p{x, w1p, b1, w2p, b2, res, out, (int)M};
    const int nres = res ? 1 : 0;
    void (*kern)(const MlpParams) = nres ? mlp_fused_kernel<1> : mlp_fused_kernel<0>;
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((__atomic_load_n(&g_mlp_attr_devs[nres], __ATOMIC_ACQUIRE) >> dev) & 1ull)) {
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS_BYTES));
        if (dev >= 0 && dev < 64) __atomic_fetch_or(&g_mlp_attr_devs[nres], 1ull << dev, __ATOMIC_RELEASE);
    }
    const unsigned blocks = (unsigned)((M + ML_BM - 1) / ML_BM);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), (size_t)ML_LDS_BYTES, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (flops_out) *flops_out = 2.0 * (double)M * 2.0 * (double)ML_C * (double)ML_HID;
    return FEMASR_OK;
}

extern "C" {

int femasr_mlp_fused(void *stream, const float *x, int64_t M, int C, int hidden, const float *w1_packed, const float *b1, const float *w2_packed,
                     const float *b2, const float *res, float *out)
{
    FEMASR_REQUIRE(femasr_mlp_fused_shape_ok(C, hidden), "mlp_fused: built for C = 256, hidden = 1024 (got %d, %d)", C, hidden);
    return femasr_mlp_fused_launch((hipStream_t)stream, x, (long long)M, w1_packed, b1, w2_packed, b2, res, out, nullptr);
}

}  // extern "C"
