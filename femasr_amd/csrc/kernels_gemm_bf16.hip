// kernels_gemm_bf16.hip — the Swin linears (network_swinir.py:19-21 fc1 / fc2, :105-107,121,143 qkv / proj) and before_quant
// (femasr_arch.py:298) as an fp32-GRADE GEMM on the bf16 matrix pipe:   out[M][N] = epi( A[M][K] . W[K][N] )
//
// Why: fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the VALU rate - 157 TFLOP/s, no co-execution with the VALU - and the
// LDS-DMA kernel of kernels_gemm.hip sits at 0.65-0.7 of that.  v_mfma_f32_32x32x16_bf16 runs 16x that rate on its own pipe.
//
// Arithmetic (restated bit for bit in oracle/femasr_oracle.c, orc_linear_bf16s):
//   * every fp32 operand is split EXACTLY into three bf16 terms,  x = x1 + x2 + x3:  x1 = bf16_rne(x), x2 = bf16_rne(x - x1),
//     x3 = (x - x1) - x2 (exactly a bf16 value: 8 + 8 + 8 significand bits); weights at pack time, activations while the A tile
//     is staged (fp32 rows in HBM, registers -> v_cvt_pk_bf16_f32 -> LDS);
//   * of the nine bf16 x bf16 partial products (each exact in fp32) the six of relative size >= 2^-16 are kept:
//         hi += a1 b1          lo += a3 b1, a1 b3, a2 b2, a2 b1, a1 b2  (in this order)
//     per 16-deep k step, k ascending; the three dropped ones are <= 2^-25 |a b| together (RNE terms: |x2| <= 2^-9 |x|,
//     |x3| <= 2^-18 |x|), i.e. below half an ulp of the product;
//   * two accumulators so that only the 16 (K = 256) large-term instructions round at the magnitude of the result;
//     out = epi((hi + lo) + bias);
//   * what one v_mfma_f32_32x32x16_bf16 does with its 16 products and the accumulator was probed on the hardware
//     (tools/ubench/mfma_bf16_probe.hip, fit_bf16_model.py): two sequential 8-product groups (k slots 0-7 = lanes 0-31, then 8-15);
//     per group the exact products are truncated toward zero to multiples of 2^(Ep-24) (Ep = largest exponent SUM of the group's
//     non-zero products) and summed exactly; that sum and the accumulator are floored (two's complement) to multiples of
//     2^max(Ep-24, Ed-32) (Ed = exponent of the accumulator) and added; the sum keeps 32 significant bits (floored) in front of one
//     round-to-nearest-even to fp32.  orc_mfma_dot8 restates exactly that, so this kernel stays bit-identical to the CPU oracle.
//
// Block = 256 threads = 4 waves of 64 x 64 outputs (2 x 2 tiles of 32 x 32, two accumulators each), 128 x 128 block tile, 72 KB of
// LDS -> two blocks per CU (one block's prologue / epilogue under the other's MFMAs).  The loop advances in 16-deep k STEPS:
//   * W planes: packed [step][n/32][plane][lane][8 bf16] (femasr_repack_k1_bf16s), copied into an LDS ring by LDS-DMA, 3 pieces of
//     1 KiB per wave and step;
//   * A: fp32 rows from global into registers two 32-deep chunks ahead, split on the VALU between the MFMAs of the current step and
//     written as three 16-byte LDS stores per 8 channels into the half of the A buffer the NEXT step reads; W ring of 4 stages; one
//     barrier per step.
// Round 6 built the other form the VERDICT asked for - the split done by the PRODUCERS (LayerNorm, the fc1 epilogue: three packed bf16
// planes, 6 bytes per value), the A side of the main loop pure LDS-DMA like the W side: 8 VALU instructions per step instead of 42, every
// output bit-identical - and measured it (profiles/r06_gemm_experiments.txt): -3 % / -5 % / -1 % per qkv / fc1 / fc2 launch stand-alone, the
// planes' extra bytes (LayerNorm writes 1.5x, fc1's plane epilogue +9 %) eat that in the network: 65.8 - 66.6 ms per step against 64.9 - 65.6
// with fp32 rows, alternating on two boxes.  The launch runs at the package power limit (constant operands instead of random ones: 15 %
// faster on the same instruction stream), so removing instructions raises the clock the chip can hold, not the throughput.  Removed again;
// git history (8a3f0b5) has the kernels, the packed-row layout and their tests.
// Every global access of the main loop is inline asm with hand-counted `s_waitcnt vmcnt(N)` waits (behind the LDS-DMA builtin the
// compiler waits for vmcnt(0) at the first LDS access and sinks the copies below the compute).  The counts are only right if the
// compiler emits NO VMEM instruction of its own between them: csrc/kernel_meta.py check_counted_waits() disassembles these kernels
// after every build and fails it otherwise (tests/test_host_logic.py::test_counted_waits_of_the_split_gemm).
#include "conv_common.h"
#include "detmath.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

struct GemmSParams {
    const float *A;
    const uint4 *W;          // femasr_repack_k1_bf16s
    const float *bias, *res1, *res2;
    float *out;
    int M, N, K, MB, NB, NT32;
    int imH, imW, Cin, tapmul;   // CONV form: input image size, channels per tap, ceil(2^20 / (Cin / 32)) (chunk -> tap without a division)
    int imHo, imWo, stride;      // ... output image size, stride (1 or 2)
};

__device__ float g_gs_zero_page[64];      // CONV form: what a tap outside the image reads (zero-initialised device memory)

constexpr int GS_A_HALF = 3 * 2 * 128;                // uint4 per step: [plane][granule][row]
constexpr int GS_A_BYTES = 2 * GS_A_HALF * 16;        // two steps
constexpr int GS_W_STAGE = 4 * 3 * 64;                // uint4 per step: [column tile][plane][lane]
constexpr int GS_W_STAGES = 4;
constexpr int GS_LDS_BYTES = GS_A_BYTES + GS_W_STAGES * GS_W_STAGE * 16;      // 24 576 + 49 152 = 73 728
static_assert(4 * TSCRATCH * 4 <= GS_LDS_BYTES, "epilogue scratch overlays the main-loop buffers");

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    const bf16x2_t r = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);      // v_cvt_pk_bf16_f32: RNE, low half = a
    return __builtin_bit_cast(unsigned, r);
}

// x = x1 + x2 + x3 exactly, two values at a time; returns the three packed pairs
__device__ __forceinline__ void split3_pair(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3)
{
    p1 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);
    p3 = cvt_pk_bf16(sa, sb);
}
__device__ __forceinline__ void split3_x8(const f32x4_t &u, const f32x4_t &v, uint4 &q1, uint4 &q2, uint4 &q3)
{
    split3_pair(u[0], u[1], q1.x, q2.x, q3.x);
    split3_pair(u[2], u[3], q1.y, q2.y, q3.y);
    split3_pair(v[0], v[1], q1.z, q2.z, q3.z);
    split3_pair(v[2], v[3], q1.w, q2.w, q3.w);
}

typedef __attribute__((address_space(3))) void *lds_vptr_s;

// three LDS-DMA copies of 1 KiB each: lane l moves 16 bytes from w + 64 i (uint4 units) to LDS byte address d0 + 1024 i + 16 l.
// (inline asm, see the header; M0 is handed back as found: it is a reserved register to the compiler)
__device__ __forceinline__ void dma3(const uint4 *w, unsigned d0)
{
    unsigned m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(m0_keep) : "s"(d0), "s"(d0 + 1024u), "s"(d0 + 2048u), "v"(w), "v"(w + 64), "v"(w + 128) : "memory");
}

// CONV = 1 (round 6): the same kernel as the implicit GEMM of a 3x3 pad-1 conv of stride 1 or 2 (the convs in FRONT of the codebook lookup:
// fema_utils.py:75,78 in the encoder's ResBlocks, femasr_arch.py:159 its stride-2 stages, network_swinir.py:465 behind every RSTB): row = output pixel (NHWC raster order),
// K = 9 Cin with k = (3 ky + kx) Cin + c - a 32-channel chunk of the main loop is a chunk of ONE tap, read from the neighbouring
// pixel's row, or from a page of zeros where the tap lies outside the image (the load is issued either way: the wait counts stay
// static).  Everything else - the split, the six products, the epilogue - is the linear layer's; oracle: conv3x3_bf16s (im2col +
// orc_linear_bf16s).
template <int ACT, int NRES, int CONV>
__global__ __launch_bounds__(256, 2) void gemm_bf16s_kernel(const GemmSParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *S = reinterpret_cast<uint4 *>(smem_raw);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB, mb = L / p.NB;
    const int m0 = mb * 128, n0 = nb * 128;
    const int h = lane >> 5, c31 = lane & 31;
    const int nsteps = p.K >> 4;

    // ---- W pieces of this wave: column tile `wave` of the block, 3 pieces (planes) per step, contiguous in the pack
    int ntile = (n0 >> 5) + wave;
    ntile = ntile < p.NT32 ? ntile : p.NT32 - 1;            // tiles past the packed matrix: clamp (never stored)
    const uint4 *srcW = p.W + (size_t)ntile * 192 + lane;
    const size_t wstep = (size_t)p.NT32 * 192;              // uint4 per step
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr_s)S);

    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }

    // one 16-deep step: twelve fragment reads (A: [plane][granule h][row], W: [column tile][plane][lane]) and 24 MFMAs:
    // lo += a3 b1, a1 b3, a2 b2, a2 b1, a1 b2 ; hi += a1 b1   (consecutive instructions on different accumulators).
    // (The order of the reads inside a step does not matter - all twelve as one burst behind the barrier was measured equal,
    // profiles/r05_gemm_bf16s_ubench.txt "step order": the partner wave of the SIMD covers the LDS round trips.)
    auto compute = [&](const uint4 *As, const int (&aoff)[2], const uint4 *Ws) {
        bf16x8 fa[3][2], fb[3][2];
        constexpr int order[3] = {2, 0, 1};              // planes in the order the products need them: a3 b1, a1 b3, a2 b2, ...
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int pa = order[q], pb = q == 0 ? 0 : (q == 1 ? 2 : 1);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[pa][rt] = __builtin_bit_cast(bf16x8, As[pa * 256 + aoff[rt]]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) fb[pb][ct] = __builtin_bit_cast(bf16x8, Ws[ct * 192 + pb * 64]);
        }
#define GS_TERM(ACC, PA, PB)                                                                                       \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                              \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                          \
            ACC[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][rt], fb[PB][ct], ACC[rt][ct], 0, 0, 0);
        GS_TERM(lo, 2, 0)
        GS_TERM(lo, 0, 2)
        GS_TERM(lo, 1, 1)
        GS_TERM(lo, 1, 0)
        GS_TERM(lo, 0, 1)
        GS_TERM(hi, 0, 0)
#undef GS_TERM
    };
    const int boff = (wn * 2) * 192 + lane;      // + ct*192 + plane*64

    {
        // ================= fp32 rows: thread -> row t >> 1, granule g2 = t & 1 of each step: item 0 = channels 8 g2 .. +7 of a chunk
        // (step 2c), item 1 = channels 16 + 8 g2 .. (step 2c + 1); two lanes cover 64 contiguous bytes of a row per load
        uint4 *Al = S;
        uint4 *Wl = reinterpret_cast<uint4 *>(smem_raw + GS_A_BYTES);
        const int g2 = t & 1, srow = t >> 1;
        int grow = m0 + srow;
        grow = grow < p.M ? grow : p.M - 1;                     // tail rows: clamp (computed, never stored)
        const float *srcA = p.A + (size_t)grow * p.K + 8 * g2;
        unsigned tapmask = 0;                                   // CONV: bit (3 ky + kx) = that neighbour of this thread's pixel lies inside the image
        if (CONV) {
            // row = OUTPUT pixel (n, oy, ox); the centre tap reads input pixel (oy * stride, ox * stride)
            const int ox = grow % p.imWo, oy = (grow / p.imWo) % p.imHo, n = grow / (p.imWo * p.imHo);
            const int px = ox * p.stride, py = oy * p.stride;
            srcA = p.A + (((size_t)n * p.imH + py) * p.imW + px) * p.Cin + 8 * g2;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int yy = py + tp / 3 - 1, xx = px + tp % 3 - 1;
                tapmask |= (yy >= 0 && yy < p.imH && xx >= 0 && xx < p.imW ? 1u : 0u) << tp;
            }
        }
        const int cpc = p.Cin >> 5;                             // CONV: 32-channel chunks per tap
        const int dstA = g2 * 128 + ((srow + 4 * g2) & 127);    // uint4 index inside a plane of a half (plane stride 256); row rotated by 4 x granule:
                                                                // the 8-lane store groups and the 16-lane read groups are both conflict-free
        f32x4_t ar[2][4] = {};          // [set = chunk parity][item 0: two float4, item 1: two float4]
        const int nch = p.K >> 5;
        auto loadA = [&](int c, f32x4_t (&r)[4]) {              // (inline asm: every wait on these registers is explicit)
            const int cl = c < nch ? c : nch - 1;
            const float *q;
            if (CONV) {
                const int tap = (cl * p.tapmul) >> 20, cc = cl - tap * cpc;           // (uniform)
                const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
                const int off = ((ky - 1) * p.imW + (kx - 1)) * p.Cin + cc * 32;
                q = ((tapmask >> tap) & 1u) ? srcA + off : g_gs_zero_page + 8 * g2;
            } else {
                q = srcA + cl * 32;
            }
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:80"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(q) : "memory");
        };
        const unsigned ldsW = lds0 + (unsigned)GS_A_BYTES + (unsigned)wave * 3072u;
        auto issueW = [&](int s) {
            dma3(srcW + (size_t)(s < nsteps ? s : nsteps - 1) * wstep, ldsW + (unsigned)(s & (GS_W_STAGES - 1)) * (GS_W_STAGE * 16));
        };
        auto splitA = [&](const f32x4_t &u, const f32x4_t &v, int half) {
            uint4 q1, q2, q3;
            split3_x8(u, v, q1, q2, q3);
            uint4 *d = Al + half * GS_A_HALF + dstA;
            d[0] = q1;
            d[256] = q2;
            d[512] = q3;
        };
        int aoff[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = wm * 64 + rt * 32 + c31;
            aoff[rt] = h * 128 + ((row + 4 * h) & 127);
        }
        auto step = [&](int s) { compute(Al + (s & 1) * GS_A_HALF, aoff, Wl + (s & (GS_W_STAGES - 1)) * GS_W_STAGE + boff); };
        // VMEM issue order (the vmcnt arithmetic below depends on it): step s issues W(s + 3) [3 copies]; an odd step 2c + 1 then issues the
        // rows of chunk c + 2 [4 loads].  The prologue plays steps -3, -2, -1.
        issueW(0);
        loadA(0, ar[0]);
        issueW(1);
        issueW(2);
        loadA(1, ar[1]);
        asm volatile("s_waitcnt vmcnt(10)" : "+v"(ar[0][0]), "+v"(ar[0][1]), "+v"(ar[0][2]), "+v"(ar[0][3]) :: "memory");      // rows of chunk 0
        splitA(ar[0][0], ar[0][1], 0);

        auto chunk = [&](int c, f32x4_t (&cur)[4], f32x4_t (&nxt)[4]) {
            // ---- step 2c: W(2c) landed (14 younger copies / loads may be in flight), last step's LDS stores done
            asm volatile("s_waitcnt vmcnt(14)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issueW(2 * c + 3);
            splitA(cur[2], cur[3], 1);                       // second half of chunk c -> read by step 2c + 1
            step(2 * c);
            // ---- step 2c + 1: W(2c + 1) landed
            asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issueW(2 * c + 4);
            loadA(c + 2, cur);                               // both items of chunk c are split: the set is free
            asm volatile("s_waitcnt vmcnt(10)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]) :: "memory");      // rows of chunk c + 1
            splitA(nxt[0], nxt[1], 0);                       // first half of chunk c + 1 -> read by step 2c + 2
            step(2 * c + 1);
        };
#pragma unroll 1
        for (int c = 0; c < nch; c += 2) {                   // (K % 64 == 0 is required by the launcher: the register sets alternate statically)
            chunk(c, ar[0], ar[1]);
            chunk(c + 1, ar[1], ar[0]);
        }
    }

    // ---- epilogue: out = act((hi + lo) + bias) + res1 + res2, in that order.  Each 32 x 32 tile goes through a per-wave LDS scratch.
    // (first: every asm load / copy of the main loop has landed - the compiler does not know that the register sets of the surplus row loads
    // are still being written and would reuse them for the addresses below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float *T = reinterpret_cast<float *>(smem_raw) + wave * TSCRATCH;
    {
        // fp32 output: lane l owns columns 4 (l & 7) .. +3 of tile rows (l >> 3) + 8 k: float4 loads / stores (the epilogue of kernels_gemm.hip).
        // Bias (two column tiles) and the residual rows of the first tile are requested BEFORE the last barrier, the residuals of tile t + 1 before
        // tile t is processed: one exposed round trip per block instead of one per tile (profiles/r05_gemm_bf16s_ubench.txt).
        const int trow = lane >> 3, tq = lane & 7;
        const float *ra = p.res1 ? p.res1 : p.res2, *rb = (p.res1 && p.res2) ? p.res2 : nullptr;
        const bool vec = (p.N & 3) == 0;
        f32x4_t bpre[2] = {}, r1[2][4] = {}, r2[2][4] = {};
        auto fetch_res = [&](int tl, f32x4_t (&d1)[4], f32x4_t (&d2)[4]) {
            if (NRES == 0 || !vec) return;
            const int rbase = m0 + (wm * 2 + (tl >> 1)) * 32, col = n0 + (wn * 2 + (tl & 1)) * 32 + 4 * tq;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = rbase + trow + 8 * k;
                const size_t o = (col < p.N && row < p.M) ? (size_t)row * p.N + col : 0;
                d1[k] = *reinterpret_cast<const f32x4_t *>(ra + o);
                if (NRES >= 2) d2[k] = *reinterpret_cast<const f32x4_t *>(rb + o);
            }
        };
        if (vec) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + (wn * 2 + j) * 32 + 4 * tq;
                bpre[j] = *reinterpret_cast<const f32x4_t *>(p.bias + (col < p.N ? col : 0));
            }
        }
        fetch_res(0, r1[0], r2[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (no copy is still writing LDS: vmcnt(0) above; the bias / residual requests stay in flight across the barrier)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const int i = tl >> 1, j = tl & 1;
            const int rbase = m0 + (wm * 2 + i) * 32, cbase = n0 + (wn * 2 + j) * 32;
            if (vec) {
                const int col = cbase + 4 * tq;
                const bool cok = col < p.N;
                if (tl + 1 < 4) fetch_res(tl + 1, r1[(tl + 1) & 1], r2[(tl + 1) & 1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * h) * TPITCH + c31] = hi[i][j][r] + lo[i][j][r];
                const f32x4_t b4 = bpre[j];
                // (same wave wrote and reads the scratch: LDS ops of one wave complete in order)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4_t a4 = *reinterpret_cast<const f32x4_t *>(T + (trow + 8 * k) * TPITCH + 4 * tq);
                    float v[4] = {a4[0] + b4[0], a4[1] + b4[1], a4[2] + b4[2], a4[3] + b4[3]};
                    if (ACT == FEMASR_ACT_GELU) {
                        const det_f32x2 g0 = det_gelu2(det_f32x2{v[0], v[1]}), g1 = det_gelu2(det_f32x2{v[2], v[3]});
                        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (NRES >= 1) v[e] = v[e] + r1[tl & 1][k][e];
                        if (NRES >= 2) v[e] = v[e] + r2[tl & 1][k][e];
                    }
                    const int row = rbase + trow + 8 * k;
                    if (cok && row < p.M) *reinterpret_cast<f32x4_t *>(p.out + (size_t)row * p.N + col) = f32x4_t{v[0], v[1], v[2], v[3]};
                }
            } else {
                const int col = cbase + c31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < p.M && col < p.N) {
                        const size_t o = (size_t)row * p.N + col;
                        float v = (hi[i][j][r] + lo[i][j][r]) + p.bias[col];
                        if (ACT == FEMASR_ACT_GELU) v = det_gelu(v);
                        if (NRES >= 1) v = v + ra[o];
                        if (NRES >= 2) v = v + rb[o];
                        p.out[o] = v;
                    }
                }
            }
        }
    }
}

// [step][n/32][plane][lane] x 8 bf16  <-  W[n][k] (torch (out, in)), zero padded in n;
// lane (j = lane & 31, h = lane >> 5) holds k = 16 step + 8 h + 0..7 of column n = 32 (n/32) + j.
__global__ void repack_k1_bf16s_kernel(const float *__restrict__ in, int O, int I, uint4 *__restrict__ out, size_t total)
{
    const int NT32 = (O + 31) / 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t rest = i >> 6;
        const int nt = (int)(rest % NT32), st = (int)(rest / NT32);
        const int n = 32 * nt + (lane & 31), k0 = 16 * st + 8 * (lane >> 5);
        f32x4_t u = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
        if (n < O) {
            u = *reinterpret_cast<const f32x4_t *>(in + (size_t)n * I + k0);
            v = *reinterpret_cast<const f32x4_t *>(in + (size_t)n * I + k0 + 4);
        }
        uint4 q1, q2, q3;
        split3_x8(u, v, q1, q2, q3);
        uint4 *o = out + ((size_t)st * NT32 + nt) * 192 + lane;
        o[0] = q1;
        o[64] = q2;
        o[128] = q3;
    }
}

// the same pack for a 3x3 conv weight (O, I, 3, 3): K = 9 I with k = (3 ky + kx) I + c
__global__ void repack_oihw_bf16s_kernel(const float *__restrict__ in, int O, int I, uint4 *__restrict__ out, size_t total)
{
    const int NT32 = (O + 31) / 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t rest = i >> 6;
        const int nt = (int)(rest % NT32), st = (int)(rest / NT32);
        const int n = 32 * nt + (lane & 31), k0 = 16 * st + 8 * (lane >> 5);
        const int tap = k0 / I, c0 = k0 - tap * I;                  // (8 consecutive channels of one tap: I % 16 == 0)
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (n < O)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = in[((size_t)n * I + c0 + j) * 9 + tap];
        uint4 q1, q2, q3;
        split3_x8(f32x4_t{v[0], v[1], v[2], v[3]}, f32x4_t{v[4], v[5], v[6], v[7]}, q1, q2, q3);
        uint4 *o = out + ((size_t)st * NT32 + nt) * 192 + lane;
        o[0] = q1;
        o[64] = q2;
        o[128] = q3;
    }
}

// test hook: case n = (a[n][16], b[n][16], c[n]) on the diagonal (n % 32, n % 32) of instruction n / 32; k slot s = 8 (lane / 32) + element
__global__ void mfma_bf16_probe_kernel(const unsigned short *__restrict__ a, const unsigned short *__restrict__ b, const float *__restrict__ c,
                                       int N, float *__restrict__ d)
{
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int n = blockIdx.x * 32 + i;
    union { bf16x8 v; unsigned short u[8]; } fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa.u[e] = n < N ? a[(size_t)n * 16 + 8 * h + e] : 0; fb.u[e] = n < N ? b[(size_t)n * 16 + 8 * h + e] : 0; }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ((r & 3) + 8 * (r >> 2) + 4 * h == i && n < N) ? c[n] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) + 4 * h == i && n < N) d[n] = acc[r];
}

struct GSVariant {
    const char *name;
    void (*kern)(const GemmSParams);
    unsigned long long attr_devs;       // bit d: MaxDynamicSharedMemorySize set on device d
};
#define GS_VARIANT(ACT, NRES) { "gemm_bf16s<act=" #ACT ",nres=" #NRES ">", gemm_bf16s_kernel<ACT, NRES, 0>, 0ull }
#define GC_VARIANT(NRES) { "conv3x3_bf16s<128px x128,9Cin split GEMM,nres=" #NRES ">", gemm_bf16s_kernel<0, NRES, 1>, 0ull }
GSVariant g_gsv[] = { GS_VARIANT(0, 0), GS_VARIANT(0, 1), GS_VARIANT(0, 2), GS_VARIANT(1, 0), GS_VARIANT(1, 1), GS_VARIANT(1, 2),
                      GC_VARIANT(0), GC_VARIANT(1), GC_VARIANT(2) };      // 6 + residual operands: the 3x3 conv form
constexpr int kNumGS = sizeof(g_gsv) / sizeof(g_gsv[0]);

}  // namespace

extern "C" size_t femasr_packed_weight_bf16s_bytes(int O, int I) { return (size_t)(I / 16) * ((O + 31) / 32) * 192 * sizeof(uint4); }
bool femasr_gemm_bf16s_shape_ok(const femasr_conv_args *a)
{
    if (a->up2 || (a->Cin % 64) != 0 || a->prologue != FEMASR_PRO_NONE) return false;
    return (a->ksz == 1 && a->pad == 0 && a->stride == 1) || femasr_conv3x3_bf16s_shape_ok(a);
}
// the 3x3 form: stride 1 or 2, pad 1, no activation; the GroupNorm + SiLU of a ResBlock conv is applied by femasr_gn_silu_apply in front of it
bool femasr_conv3x3_bf16s_shape_ok(const femasr_conv_args *a)
{
    return a->ksz == 3 && (a->stride == 1 || a->stride == 2) && a->pad == 1 && !a->up2 && (a->Cin % 64) == 0 && a->Cin <= 1024 && a->prologue == FEMASR_PRO_NONE &&
           a->act == FEMASR_ACT_NONE && (long long)a->B * a->H * a->W * a->Cin < (1ll << 31);
}
int femasr_gemm_bf16s_variant_count() { return kNumGS; }
const char *femasr_gemm_bf16s_variant_name(int v) { return (v >= 0 && v < kNumGS) ? g_gsv[v].name : "?"; }

extern "C" int femasr_repack_k1_bf16s(void *stream, const float *in, int O, int I, void *out)
{
    hipStream_t s = (hipStream_t)stream;
    FEMASR_REQUIRE(in && out && O > 0 && I > 0 && (I % 64) == 0, "repack_k1_bf16s: bad args");
    const size_t total = (size_t)(I / 16) * ((O + 31) / 32) * 64;
    size_t g = (total + 255) / 256;
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(repack_k1_bf16s_kernel, dim3((unsigned)g), dim3(256), 0, s, in, O, I, (uint4 *)out, total);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

extern "C" size_t femasr_packed_weight_conv3x3_bf16s_bytes(int O, int I) { return femasr_packed_weight_bf16s_bytes(O, 9 * I); }
extern "C" int femasr_repack_oihw_bf16s(void *stream, const float *in, int O, int I, void *out)
{
    FEMASR_REQUIRE(in && out && O > 0 && I > 0 && (I % 64) == 0, "repack_oihw_bf16s: needs a 3x3 OIHW weight with I %% 64 == 0");
    const size_t total = (size_t)(9 * I / 16) * ((O + 31) / 32) * 64;
    size_t g = (total + 255) / 256;
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(repack_oihw_bf16s_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, O, I, (uint4 *)out, total);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

extern "C" int femasr_debug_mfma_bf16(void *stream, const uint16_t *a, const uint16_t *b, const float *c, int n, float *d)
{
    FEMASR_REQUIRE(a && b && c && d && n > 0, "debug_mfma_bf16: bad args");
    hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, (hipStream_t)stream, a, b, c, n, d);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

int femasr_gemm_bf16s_launch(hipStream_t s, const femasr_conv_args *a, const void *w_bf16s, int *variant_out, double *flops_out)
{
    FEMASR_REQUIRE(a && a->in && w_bf16s && femasr_gemm_bf16s_shape_ok(a), "gemm_bf16s: layer is neither a 1x1 / linear layer nor a 3x3 stride-1 pad-1 conv with Cin %% 64 == 0");
    FEMASR_REQUIRE(a->bias && a->out, "gemm_bf16s: bias/out must be set");
    FEMASR_REQUIRE(a->act == FEMASR_ACT_NONE || a->act == FEMASR_ACT_GELU, "gemm_bf16s: bad activation %d", a->act);
    const bool conv = a->ksz == 3;
    const int Ho = conv ? (a->H - 1) / a->stride + 1 : a->H, Wo = conv ? (a->W - 1) / a->stride + 1 : a->W;      // (H + 2 - 3) / stride + 1
    const long long M = (long long)a->B * Ho * Wo;
    FEMASR_REQUIRE(a->Ho == Ho && a->Wo == Wo, "gemm_bf16s: Ho/Wo mismatch");
    FEMASR_REQUIRE(M > 0 && M < (1ll << 31) - 256, "gemm_bf16s: bad row count");
    GemmSParams p{};
    p.A = a->in; p.W = (const uint4 *)w_bf16s; p.bias = a->bias; p.res1 = a->res1; p.res2 = a->res2; p.out = a->out;
    p.M = (int)M; p.N = a->Cout; p.K = conv ? 9 * a->Cin : a->Cin;
    p.imH = a->H; p.imW = a->W; p.imHo = Ho; p.imWo = Wo; p.stride = a->stride; p.Cin = a->Cin; p.tapmul = ((1 << 20) + (a->Cin >> 5) - 1) / (a->Cin >> 5);
    p.NT32 = (p.N + 31) / 32;
    p.MB = (p.M + 127) / 128; p.NB = (p.N + 127) / 128;
    const int nres = (a->res1 ? 1 : 0) + (a->res2 ? 1 : 0), act = a->act == FEMASR_ACT_GELU ? 1 : 0;
    const int vi = conv ? 6 + nres : 3 * act + nres;
    GSVariant &v = g_gsv[vi];
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((__atomic_load_n(&v.attr_devs, __ATOMIC_ACQUIRE) >> dev) & 1ull)) {      // (idempotent: a race only repeats the call)
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_BYTES));
        if (dev >= 0 && dev < 64) __atomic_fetch_or(&v.attr_devs, 1ull << dev, __ATOMIC_RELEASE);
    }
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(p.MB * p.NB)), dim3(256), (size_t)GS_LDS_BYTES, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (variant_out) *variant_out = vi;
    if (flops_out) *flops_out = 2.0 * (double)M * (double)a->Cout * (double)p.K;
    return FEMASR_OK;
}
