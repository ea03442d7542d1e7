// kernels_wino_up2.hip — nn.Upsample(x2, nearest) + 3x3 stride-1 pad-1 conv (femasr_arch.py:172-173,202-203) in a
// Winograd-type minimal form on the fp32 matrix cores: 25 multiplies per 4x4 outputs.
//
// A 4x4 tile of the OUTPUT (upsampled grid) reads a 6x6 patch of the upsampled image, i.e. only a 4x4 patch of the
// low-resolution input: per dimension d = (l0, l1, l1, l2, l2, l3).  Put through the F(4,3) input transform B^T that gives
//     r0 = 4 l0 - 5 l1 + l2,   r1 = 2 (l2 - 4 l1),   r2 = 0,   r3 = 3 (l2 - l1),   r4 = -(l2 - l1),   r5 = 4 l1 - 5 l2 + l3
// - one component vanishes and two are proportional, so per dimension FIVE products are enough for four outputs:
//     v0 = fma(4, l0, fma(-5, l1, l2))   v1 = fma(-4, l1, l2)   v3 = l2 - l1   v5 = fma(4, l1, fma(-5, l2, l3))
//     e0 = g0/4   e1 = -((g0 + g1) + g2)/3   eP = fma(4, g2, fma(4, g1, g0))/12   eQ = fma(4, g2, g0 + g1)/6   e5 = g2
//     m0 = e0 v0,  m1 = e1 v1,  mP = eP v3,  mQ = eQ v3,  m5 = e5 v5          (each summed over the input channels)
//     y0 = (m0 + m1) + mP    y1 = fma(2, mQ, m1)    y2 = fma(4, mP, m1)    y3 = fma(8, mQ, m1) + m5
// (the constants 2, 3 and -1 of r1, r3, r4 are folded into the filters).  In 2-D: 25 products per 16 outputs and channel
// pair where the phase-filter form of kernels_conv.hip needs 64 and the definition 144.  Like the F(4x4,3x3) kernel this is
// only used behind the codebook lookup of single-codebook networks; all fp32, every value one fixed sequence of IEEE
// operations restated by oracle/femasr_oracle.c orc_conv_up2_winograd.
//
// Structure = kernels_wino.hip (read its header first): block = 8 waves, two 16x16-pixel OUTPUT sub-blocks (= 2 x 16 tiles =
// one 32-row MFMA tile) x 64 output channels, K in steps of 8 input channels, every wave M phase then T phase, one barrier
// per step.  What differs:
//   staging   the 10x10 low-resolution patch of a sub-block (8x8 + halo), plain copy (these convs have no GN prologue)
//   T phase   thread = (tile, channel, half): 12 LDS reads -> 8 of the 16 DISTINCT transformed values V[4][4] (components P and
//             Q share v3): V[16][32 tiles][8 channels]
//   M phase   50 (component, 32-column tile) pairs: wave w owns column tile w&1 of components w/2 + 4q, q = 0..5 (waves 0, 1
//             also q = 6): 13 / 13 / 12 / 12 pairs per SIMD
//   epilogue  Mx[25][32][32] per column tile, thread = (tile, channel) applies the 5 -> 4 output transform twice
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float tf2 __attribute__((ext_vector_type(2)));

#ifdef FEMASR_WINO_TT      // tools/build_debug.sh tt: cycle stamps of waves 0 and 7 of every block in a global buffer, see kernels_wino.hip
// (same slots: 63 start, 0 prologue done, 16 + s step s done, 1 main loop done, 2..5 + 4r epilogue round r, 10 end)
__device__ unsigned long long *g_wu_ttbuf;
#define WUTT(slot) { if (lane == 0 && (wave == 0 || wave == 7)) g_wu_ttbuf[((size_t)blockIdx.x * 2 + (wave == 7 ? 1 : 0)) * 64 + (slot)] = __builtin_readcyclecounter(); }
#define WUTTR(slot) { if (lane == 0 && (wave == 0 || wave == 7)) g_wu_ttbuf[((size_t)blockIdx.x * 2 + (wave == 7 ? 1 : 0)) * 64 + (slot)] = wall_clock64(); }
#define WUTT_INIT { WUTTR(62) WUTT(63) }
#define WUTT_END { WUTT(10) WUTTR(11) }
#else
#define WUTT(slot) {}
#define WUTT_INIT
#define WUTT_END {}
#endif

// (Measured in round 4 and removed: the transform's patch reads at the start of the M phase, or the whole transform inside the M phase -
// +-1 %, profiles/r04_wino_variants.txt.  Kept: FEMASR_WINO_DEEP, the two-set patch prefetch of kernels_wino.hip.)
#ifndef FEMASR_WINO_DEEP
#define FEMASR_WINO_DEEP 1
#endif
#ifndef FEMASR_WINO_NT       // experiment: cache-policy bits of the streaming accesses (bit 1 = nt): 1 = input patches, 2 = residual loads / output stores
#define FEMASR_WINO_NT 0
#endif
#define W_NT_IN ((FEMASR_WINO_NT & 1) ? 2 : 0)
#define W_NT_IO ((FEMASR_WINO_NT & 2) ? 2 : 0)

namespace {

struct WinoUpParams {
    const float *in, *in2, *u, *bias, *res1, *res2;      // in2: optional second input, added to `in` while staging (femasr_conv_args.in_add)
    float *out;
    double *gn_part;
    int B, H, W, Cin, Cout, Ho, Wo;     // H, W: low-resolution input; Ho = 2H, Wo = 2W
    int sbX, sbY, nsb;                  // 16x16-pixel output sub-blocks per row / per column / in the batch
    int MB, NB, nsteps, NT32;
};

constexpr int WU_NT = 512;
constexpr int WU_PS = 12;                         // floats per patch pixel: 8 channels + 4
constexpr int WU_PW = 10;
constexpr int WU_PPIX = WU_PW * WU_PW;            // 100
constexpr int WU_PSZ = 2 * WU_PPIX * WU_PS;       // floats per patch buffer (two sub-blocks): 2400
constexpr int WU_VSZ = 16 * 32 * 8;               // floats per V buffer: 4096
constexpr int WU_MAIN = 2 * WU_PSZ + 2 * WU_VSZ;
constexpr int WU_MX = 25 * 32 * 32;               // epilogue: one 32-column tile of all components
constexpr int WU_RED = 8 * 2 * 16 * 2 * 2;        // floats: [8 waves][2 sub-blocks][<= 16 groups][2] doubles
constexpr int WU_UNITS = 2 * WU_PPIX * 2;         // float4 staging units per step: 400

inline size_t wino_up_lds_bytes() { return (size_t)((WU_MX + WU_RED) > WU_MAIN ? (WU_MX + WU_RED) : WU_MAIN) * sizeof(float); }

// output rows of the 5 -> 4 transform (see the header), on PAIRS (two horizontally adjacent tiles): v_pk_add_f32 / v_pk_fma_f32
// are IEEE per component
__device__ __forceinline__ void at5(tf2 m0, tf2 m1, tf2 mP, tf2 mQ, tf2 m5, tf2 &y0, tf2 &y1, tf2 &y2, tf2 &y3)
{
    y0 = (m0 + m1) + mP;
    y1 = __builtin_elementwise_fma(tf2{2.0f, 2.0f}, mQ, m1);
    y2 = __builtin_elementwise_fma(tf2{4.0f, 4.0f}, mP, m1);
    y3 = __builtin_elementwise_fma(tf2{8.0f, 8.0f}, mQ, m1) + m5;
}

template <int NRES, bool ADD>
__global__ __launch_bounds__(WU_NT, 2) void conv3x3_wino_up2_kernel(const WinoUpParams p)
{
    constexpr bool HAS1 = NRES >= 1, HAS2 = NRES >= 2;      // residual operands of the epilogue (compile time; the network's x2 convs have none)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                        // [2][WU_PSZ]
    float *Vs = smem + 2 * WU_PSZ;           // [2][WU_VSZ]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    WUTT_INIT
    const int c31 = lane & 31, hh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB, mb = L / p.NB;
    const int n0 = nb * 64;

    // ---- the two output sub-blocks (uniform)
    int sn[2], sy0[2], sx0[2], sbi[2];
    bool sval[2];
#pragma unroll
    for (int z = 0; z < 2; ++z) {
        const int sb = 2 * mb + z, per = p.sbY * p.sbX;
        sval[z] = sb < p.nsb;
        const int sbc = sval[z] ? sb : 0;
        sn[z] = sbc / per;
        sbi[z] = sbc - sn[z] * per;
        const int by = sbi[z] / p.sbX, bx = sbi[z] - by * p.sbX;
        sy0[z] = 16 * by;
        sx0[z] = 16 * bx;
    }

    // ---- staging unit of this thread (threads 0..399): (sub-block z, pixel of its 10x10 low-resolution patch, channel quad)
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void *)(p.in + (size_t)sn[0] * p.H * p.W * p.Cin), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, 0x7fffffff, 0x00020000);
    unsigned goff = 0xffffffffu;             // beyond num_records: the load returns zeros (padding, invalid sub-block, idle threads)
    int sdst = 0;
    {
        const int z = t >= 2 * WU_PPIX ? 1 : 0, r = t - 2 * WU_PPIX * z;
        const int pix = r >> 1, quad = r & 1;
        const int py = pix / WU_PW, px = pix - py * WU_PW;
        const int y = ((z ? sy0[1] : sy0[0]) >> 1) - 1 + py, x = ((z ? sx0[1] : sx0[0]) >> 1) - 1 + px;
        const bool ok = t < WU_UNITS && (z ? sval[1] : sval[0]) && y >= 0 && y < p.H && x >= 0 && x < p.W;
        if (ok) goff = (unsigned)((((size_t)(z ? sn[1] - sn[0] : 0) * p.H + y) * p.W + x) * p.Cin + 4 * quad) * 4u;
        sdst = (z * WU_PPIX + pix) * WU_PS + 4 * quad;
    }
    constexpr bool DEEP = FEMASR_WINO_DEEP != 0;      // two register sets: the patch requested in step s is staged in step s+1 (kernels_wino.hip)
    // ADD: the conv reads in + in2 (the decoder's `x + enc_feats[i]`, femasr_arch.py:361-362): a second request with the same offsets, one
    // packed add pair per unit while staging (zero padding stays zero: both requests return 0 there)
    const __amdgpu_buffer_rsrc_t rsrc_in2 = __builtin_amdgcn_make_buffer_rsrc((void *)((ADD ? p.in2 : p.in) + (size_t)sn[0] * p.H * p.W * p.Cin), 0, 0x7fffffff, 0x00020000);
    struct Unit { f32x4_t a, b; };
    Unit rp, rq;
    auto load_patch_to = [&](Unit &rr, int s) {       // unconditional (steps past the end re-read the last one): the wait counters stay static
        const int sc = s < p.nsteps ? s : p.nsteps - 1;
        rr.a = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, goff, sc * 32, W_NT_IN));
        if (ADD) rr.b = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in2, goff, sc * 32, W_NT_IN));
    };
    auto load_patch = [&](int s) { load_patch_to(rp, s); };
    auto store_patch_from = [&](const Unit &rr, int buf) {
        if (t < WU_UNITS) *reinterpret_cast<f32x4_t *>(Ps + buf * WU_PSZ + sdst) = ADD ? rr.a + rr.b : rr.a;
    };
    auto store_patch = [&](int buf) { store_patch_from(rp, buf); };

    // ---- input transform item: (tile tm, channel tch); waves 0-3 produce the rows v0, v1 of the column pass, waves 4-7 v3, v5
    // (w and w+4 share a SIMD: each SIMD carries both halves)
    const int thalf = wave >> 2;
    const int titem = (wave & 3) * 64 + lane;
    const int tm = titem >> 3, tch = titem & 7;
    // patch rows 2 ety + thalf .. + 2, columns 2 etx .. 2 etx + 3 of sub-block tm >> 4
    const int tsrc = (tm >> 4) * WU_PPIX * WU_PS + ((2 * ((tm & 15) >> 2) + thalf) * WU_PW + 2 * (tm & 3)) * WU_PS + tch;
    const int tdst = thalf * 8 * 256 + tm * 8 + (tch & 1) * 4 + (tch >> 1);
    float td[3][4];
    auto transform_read = [&](int pbuf) {
        const float *src = Ps + pbuf * WU_PSZ + tsrc;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) td[a][b] = src[(a * WU_PW + b) * WU_PS];
    };
    auto transform_write = [&](int vbuf) {
        float *dst = Vs + vbuf * WU_VSZ + tdst;
        float r[2][4];
        if (thalf == 0) {            // rows l0, l1, l2 -> v0, v1
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                r[0][b] = __builtin_fmaf(4.0f, td[0][b], __builtin_fmaf(-5.0f, td[1][b], td[2][b]));
                r[1][b] = __builtin_fmaf(-4.0f, td[1][b], td[2][b]);
            }
        } else {                     // rows l1, l2, l3 -> v3, v5
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                r[0][b] = td[1][b] - td[0][b];
                r[1][b] = __builtin_fmaf(4.0f, td[0][b], __builtin_fmaf(-5.0f, td[1][b], td[2][b]));
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *o = dst + i * 4 * 256;
            o[0 * 256] = __builtin_fmaf(4.0f, r[i][0], __builtin_fmaf(-5.0f, r[i][1], r[i][2]));
            o[1 * 256] = __builtin_fmaf(-4.0f, r[i][1], r[i][2]);
            o[2 * 256] = r[i][2] - r[i][1];
            o[3 * 256] = __builtin_fmaf(4.0f, r[i][1], __builtin_fmaf(-5.0f, r[i][2], r[i][3]));
        }
    };

    // ---- M phase: pair q of wave w = (component w/2 + 4 q, column tile w & 1); q = 6 only exists for waves 0 and 1
    f32x16 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned lw = (unsigned)lane * 16u;
    const int aoff = c31 * 8 + hh * 4;
    const int wcomp0 = wave >> 1, wnt = wave & 1;
    const bool has6 = wave < 2;
    auto pcomp = [&](int q) -> int { const int c = wcomp0 + 4 * q; return c < 25 ? c : 24; };             // (clamped: the surplus load of waves 2-7 is never used)
    auto vcomp = [&](int q) -> int {           // component (i, j) of [0, 1, P, Q, 5]^2 -> distinct transformed value (P and Q share v3)
        const int c = pcomp(q), i = c / 5, j = c - 5 * i;
        return ((i > 2 ? i - 1 : i) << 2) + (j > 2 ? j - 1 : j);
    };
    auto ldU = [&](int s, int q) -> f32x4_t {  // unconditional, like load_patch
        const int sc = s < p.nsteps ? s : p.nsteps - 1;
        return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, lw, (((sc * 25 + pcomp(q)) * p.NT32 + 2 * nb + wnt) << 10), 0));
    };
    // U fragments: ring = pairs 0-2 of the NEXT step (loaded behind this step's pairs 0-2), early = pairs 3-5 (issued at the end of
    // the T phase, in flight across the barrier), late = pair 6 (issued at the start of the M phase)
    f32x4_t ring[3], early[3], late;
    auto issue_early = [&](int s) {
#pragma unroll
        for (int q = 3; q < 6; ++q) early[q - 3] = ldU(s, q);
    };
    auto mphase = [&](int s, auto par_c) {      // par_c = s & 1 at compile time
        constexpr int PAR = decltype(par_c)::value;
        const float *Vb = Vs + PAR * WU_VSZ + aoff;
        late = ldU(s, 6);
        // A fragments: the one of pair q+1 is REQUESTED before the MFMAs of pair q (two register sets, pinned with sched_barrier:
        // left alone the scheduler reuses one set and sinks the read below the MFMAs - an LDS round trip exposed per pair)
        f32x4_t af[2];
        af[0] = *reinterpret_cast<const f32x4_t *>(Vb + vcomp(0) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            if (q + 1 < 7) af[(q + 1) & 1] = *reinterpret_cast<const f32x4_t *>(Vb + vcomp(q + 1) * 256);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4_t a = af[q & 1];
            const f32x4_t b = q < 3 ? ring[q] : (q < 6 ? early[q - 3] : late);
            if (q < 6 || has6) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[q], 0, 0, 0);
            }
            if (q < 3) ring[q] = ldU(s + 1, q);
            if (q == 4) {      // behind the step's U requests (loads return in order)
                if (!DEEP) load_patch(s + 2);                     // staged in THIS step's T phase
                else if (PAR) load_patch_to(rp, s + 3);           // staged in the NEXT step's T phase, from the other set
                else load_patch_to(rq, s + 3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: the patches of steps 0 and 1 requested together (two register sets), then - DEEP - the one step 0 stages
    load_patch(0);
    load_patch_to(rq, 1);
#pragma unroll
    for (int q = 0; q < 3; ++q) ring[q] = ldU(0, q);
    store_patch(0);
    if (DEEP) load_patch(2);
    store_patch_from(rq, 1);
    __syncthreads();
    transform_read(0);
    transform_write(0);
    issue_early(0);
    __syncthreads();
    WUTT(0)

    // ---- main loop.  Per step: M(s); transform of step s+1 (its patch was staged during step s-1); stage step s+2.
    // (Measured and dropped: 32-channel blocks with half the accumulators, <= 128 VGPRs and a 16-column epilogue exchange, so that
    // TWO blocks are resident per CU and one's prologue / epilogue runs under the other's MFMAs - bit-exact, and exactly as fast
    // as this form: 1.68 / 1.91 ms on both layer shapes.  The block's ends are instruction issue, not idle latency.
    // Also measured and dropped: the two waves of a SIMD taking the phases in opposite order - waves 0-3 M then T, waves 4-7 T then
    // M - so that one wave's LDS round trips run under its partner's MFMAs: 2.10 vs 1.68 ms on 256 -> 128 channels; the merged
    // control flow also costs a spilled accumulator tile per step.)
    auto step = [&](int s, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        mphase(s, par_c);
        transform_read(PAR ^ 1);              // (unconditional: after the last step it transforms a stale patch into a dead buffer)
        __builtin_amdgcn_sched_barrier(0);
        transform_write(PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // step s+2 -> the buffer the transform of step s read a barrier ago (DEEP: from the set of this parity, requested a step ago)
        if (DEEP && PAR) store_patch_from(rq, PAR); else store_patch_from(rp, PAR);
        issue_early(s + 1);
        __syncthreads();
        if (s < 40) WUTT(16 + s)
    };
    for (int s = 0; s < p.nsteps; s += 2) {       // (nsteps = Cin / 8 is a multiple of 4)
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
    }
    WUTT(1)

    // ---------------------------------------------------------------------------------------------------------------
    // epilogue = the F(4x4,3x3) kernel's (kernels_wino.hip), with 25 components and the 5 -> 4 transform: one 32-column tile
    // (round) at a time, accumulators -> Mx[component][tile pair][channel][2] (rows e, e + 1 of an accumulator tile = two
    // horizontally adjacent tiles = one aligned register pair = one ds_write_b64); a thread = (tile pair, channel) runs both tiles
    // as packed fp32; outputs / residuals through buffer instructions with one per-lane offset; nothing is masked in a block
    // whose 32 tiles lie inside the image (FULL); GroupNorm partial moments in the order of orc_gn_wino_partial.
    float *Mx = smem;
    double *red = reinterpret_cast<double *>(smem + WU_MX);
    const bool gnp = p.gn_part != nullptr;
    const int cg = p.Cout >> 5;                      // channels per GroupNorm group (>= 2: Cout % 64 == 0)
    const int gpt = 32 / (cg < 32 ? cg : 32);        // groups per 32-channel tile (<= 16)
    const int pi = t >> 5;                           // tile pair: rows 2 pi, 2 pi + 1 of the accumulator tiles
    const int ez = wave >> 2, ety = wave & 3, etx = 2 * hh;      // sub-block (uniform), tile row (uniform), left tile of the pair
    unsigned vmask[2] = {0xffffu, 0xffffu}, ooff;
    bool full = true;
#pragma unroll
    for (int z = 0; z < 2; ++z) full = full && (z ? sval[1] : sval[0]) && (z ? sy0[1] : sy0[0]) + 16 <= p.Ho && (z ? sx0[1] : sx0[0]) + 16 <= p.Wo;      // (uniform)
    {
        const int oy = (ez ? sy0[1] : sy0[0]) + 4 * ety, ox = (ez ? sx0[1] : sx0[0]) + 4 * etx;
        if (!full) {         // per-pixel validity of the thread's two tiles (bit 4a + b): only a block that touches the image border needs it
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                unsigned m = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) m |= ((ez ? sval[1] : sval[0]) && oy + (k >> 2) < p.Ho && ox + 4 * e + (k & 3) < p.Wo ? 1u : 0u) << k;
                vmask[e] = m;
            }
        }
        ooff = (unsigned)((((ez ? sn[1] - sn[0] : 0) * p.Ho + oy) * p.Wo + ox) * p.Cout + n0 + c31) * 4u;      // (< 2^30: two images of < 2^27 elements)
    }
    const size_t img0 = (size_t)sn[0] * p.Ho * p.Wo * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out + img0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc((void *)((HAS1 ? p.res1 : p.out) + img0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r2 = __builtin_amdgcn_make_buffer_rsrc((void *)((HAS2 ? p.res2 : p.out) + img0), 0, 0x7fffffff, 0x00020000);
    // uniform byte offset of pixel k = 4a + b of tile e, round r; the two strides are made opaque once per round (kernels_wino.hip)
    int cs_u = 0, rs_u = 0;
    auto soff = [&](int k, int e, int r) -> int { return (k >> 2) * rs_u + (4 * e + (k & 3)) * cs_u + 128 * r; };
    auto oob = [&](int e, int k) -> unsigned { return ~(unsigned)((int)(vmask[e] << (31 - k)) >> 31); };      // all-ones: pixel k of tile e is outside
    auto voff = [&](auto fullc, int e, int k) -> unsigned {
        if (decltype(fullc)::value) return ooff;
        return ooff | oob(e, k);
    };
    auto fetch = [&](auto fullc, const __amdgpu_buffer_rsrc_t rs, int r, tf2 (&dst)[16], int k0, int k1) {       // pixels k0 .. k1-1, no waits between
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < k0 || k >= k1) continue;
            dst[k][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff(fullc, 0, k), soff(k, 0, r), W_NT_IO));
            dst[k][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff(fullc, 1, k), soff(k, 1, r), W_NT_IO));
        }
    };
    auto round = [&](auto fullc, int r) {
        constexpr bool FULL = decltype(fullc)::value;
        tf2 r1[16], r2[16];
        if (!FULL) asm volatile("" : "+v"(vmask[0]), "+v"(vmask[1]));      // (opaque: the per-pixel masks are not to be hoisted out of the round loop)
        cs_u = __builtin_amdgcn_readfirstlane(p.Cout * 4);
        rs_u = __builtin_amdgcn_readfirstlane(p.Wo * p.Cout * 4);
        asm volatile("" : "+s"(cs_u), "+s"(rs_u));
        if (FULL) {          // (a border block fetches row by row below: kernels_wino.hip)
            if (HAS1) fetch(fullc, rs_r1, r, r1, 0, HAS2 ? 8 : 16);      // (two operands: upper tile rows now, lower ones behind the first pass - kernels_wino.hip)
            if (HAS2) fetch(fullc, rs_r2, r, r2, 0, 8);
        }
        const float bv = p.bias[n0 + 32 * r + c31];
        __syncthreads();
        WUTT(3 + 4 * r)
        {
            // component stride: 16 tile pairs x 32 channels = 4 KiB; two bases keep every read inside the 64-KiB offset field
            int so1 = (16 * 512 + pi * 32 + c31) * 8;
            asm volatile("" : "+v"(so1));
            const tf2 *src0 = reinterpret_cast<const tf2 *>(Mx) + pi * 32 + c31;
            const tf2 *src1 = reinterpret_cast<const tf2 *>(reinterpret_cast<const char *>(Mx) + so1);
            auto mx = [&](int c) -> tf2 { return c < 16 ? src0[c * 512] : src1[(c - 16) * 512]; };
            tf2 tt[4][5];
#pragma unroll
            for (int j = 0; j < 5; ++j)
                at5(mx(0 * 5 + j), mx(1 * 5 + j), mx(2 * 5 + j), mx(3 * 5 + j), mx(4 * 5 + j), tt[0][j], tt[1][j], tt[2][j], tt[3][j]);
            if (FULL && HAS2) { fetch(fullc, rs_r1, r, r1, 8, 16); fetch(fullc, rs_r2, r, r2, 8, 16); }
            const tf2 bv2 = {bv, bv};
            tf2 s2 = {0.f, 0.f}, ss2 = {0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                tf2 y[4];
                if (!FULL) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (HAS1) fetch(fullc, rs_r1, r, r1, 4 * a, 4 * a + 4);
                    if (HAS2) fetch(fullc, rs_r2, r, r2, 4 * a, 4 * a + 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                at5(tt[a][0], tt[a][1], tt[a][2], tt[a][3], tt[a][4], y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = 4 * a + b;
                    tf2 v = y[b] + bv2;
                    if (HAS1) v = v + r1[k];
                    if (HAS2) v = v + r2[k];
#ifdef FEMASR_WUP_NOSTORE          // experiment: everything but the output stores (tools/build_debug.sh)
                    if (p.nsteps < 0)
#endif
                    {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), rs_out, voff(fullc, 0, k), soff(k, 0, r), W_NT_IO);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1]), rs_out, voff(fullc, 1, k), soff(k, 1, r), W_NT_IO);
                    }
                    if (gnp) {
                        if (!FULL) {
                            v[0] = __uint_as_float(__float_as_uint(v[0]) & ~oob(0, k));
                            v[1] = __uint_as_float(__float_as_uint(v[1]) & ~oob(1, k));
                        }
                        s2 = s2 + v;
                        ss2 = __builtin_elementwise_fma(v, v, ss2);
                    }
                }
            }
            if (gnp) {      // fp64 from here: the two tiles of the pair, the channels of the group (xor butterfly), the two pairs of the tile row
                double gs = (double)s2[0] + (double)s2[1], gss = (double)ss2[0] + (double)ss2[1];
                for (int d = 1; d < cg && d < 32; d <<= 1) {
                    gs = gs + __shfl_xor(gs, d, 64);
                    gss = gss + __shfl_xor(gss, d, 64);
                }
                const double a2 = gs + __shfl_xor(gs, 32, 64), b2 = gss + __shfl_xor(gss, 32, 64);
                if (lane < 32 && (c31 & (cg - 1)) == 0) {
                    double *dst = red + ((size_t)wave * 16 + c31 / cg) * 2;
                    dst[0] = a2;
                    dst[1] = b2;
                }
            }
        }
        WUTT(4 + 4 * r)
    };
    // (one per-lane base, pinned as an integer offset: a wave's pairs are the components wave / 2 + 4 q; everything else is an
    // immediate offset)
    int wo0 = wcomp0 * 4096 + hh * 512 + c31 * 8;
    asm volatile("" : "+v"(wo0));
    char *wb0 = reinterpret_cast<char *>(Mx) + wo0;
    auto write_acc = [&]() {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            if (q < 6 || has6) {
                char *dst = wb0 + q * 4 * 4096;
#pragma unroll
                for (int e = 0; e < 16; e += 2)      // pair row ((e & 3) + 8 (e >> 2) + 4 hh) / 2, 256 bytes each
                    *reinterpret_cast<tf2 *>(dst + (((e & 3) >> 1) + 4 * (e >> 2)) * 256) = tf2{acc[q][e], acc[q][e + 1]};
            }
        }
    };
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        if (wnt == r) write_acc();
        WUTT(2 + 4 * r)
        if (full) round(std::true_type{}, r); else round(std::false_type{}, r);
        __syncthreads();
        WUTT(5 + 4 * r)
        if (gnp && t < 2 * gpt) {
            const int z = t / gpt, gl = t - z * gpt;
            if (z ? sval[1] : sval[0]) {
                double S = red[((size_t)(4 * z) * 16 + gl) * 2], SS = red[((size_t)(4 * z) * 16 + gl) * 2 + 1];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    S = S + red[((size_t)(4 * z + w) * 16 + gl) * 2];
                    SS = SS + red[((size_t)(4 * z + w) * 16 + gl) * 2 + 1];
                }
                const int g = (n0 + 32 * r) / cg + gl;
                double *dst = p.gn_part + (((size_t)(z ? sn[1] : sn[0]) * p.sbY * p.sbX + (z ? sbi[1] : sbi[0])) * 32 + g) * 2;
                dst[0] = S;
                dst[1] = SS;
            }
        }
    }
    WUTT_END
}

// the five 1-D filters of the form (header): rows (over ky) then columns (over kx)
__device__ __forceinline__ float e5(int r, float g0, float g1, float g2)
{
    const float c3 = -1.0f / 3.0f, c12 = 1.0f / 12.0f, c6 = 1.0f / 6.0f;
    switch (r) {
    case 0: return g0 * 0.25f;
    case 1: return ((g0 + g1) + g2) * c3;
    case 2: return __builtin_fmaf(4.0f, g2, __builtin_fmaf(4.0f, g1, g0)) * c12;
    case 3: return __builtin_fmaf(4.0f, g2, g0 + g1) * c6;
    default: return g2;
    }
}

// 3x3 OIHW -> out[step = ci/8][component 5 i + j][32-column tile][lane][e]: column o = 32 tile + lane%32,
// ci = 8 step + 2 e + lane/32 (the B fragments of the four k-pairs of a step: one 16-byte load per lane, 1 KiB per wave)
__global__ void repack_wino_up2_kernel(const float *__restrict__ in, int O, int I, float *__restrict__ out, size_t total)
{
    const int NT32 = (O + 31) / 32;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        size_t rest = idx >> 8;
        const int ntile = (int)(rest % NT32);
        rest /= NT32;
        const int comp = (int)(rest % 25), step = (int)(rest / 25);
        const int ci = 8 * step + 2 * e + (lane >> 5), o = ntile * 32 + (lane & 31);
        float v = 0.f;
        if (ci < I && o < O) {
            const int i = comp / 5, j = comp - 5 * i;
            const float *gw = in + ((size_t)o * I + ci) * 9;
            float ur[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) ur[x] = e5(i, gw[x], gw[3 + x], gw[6 + x]);
            v = e5(j, ur[0], ur[1], ur[2]);
        }
        out[idx] = v;
    }
}

typedef void (*wu_kern_t)(const WinoUpParams);
wu_kern_t g_wu_kern[6] = {conv3x3_wino_up2_kernel<0, false>, conv3x3_wino_up2_kernel<1, false>, conv3x3_wino_up2_kernel<2, false>,      // [2 * ... ]: by residual
                          conv3x3_wino_up2_kernel<0, true>, conv3x3_wino_up2_kernel<1, true>, conv3x3_wino_up2_kernel<2, true>};          // operands, then + second input
unsigned long long g_attr_devs[6] = {0, 0, 0, 0, 0, 0};

}  // namespace

bool femasr_conv_wino_up2_shape_ok_lim(const femasr_conv_args *a, int log2_total, int log2_image)
{
    const size_t tot = (size_t)1 << log2_total, img = (size_t)1 << log2_image;
    return a->ksz == 3 && a->stride == 1 && a->pad == 1 && a->up2 && a->act == FEMASR_ACT_NONE && a->prologue == FEMASR_PRO_NONE &&
           (a->Cin % BK) == 0 && a->Cin <= 1024 && (a->Cout % 64) == 0 &&
           (size_t)a->B * a->H * a->W * a->Cin < tot && (size_t)a->B * 4 * a->H * a->W * a->Cout < tot &&
           (size_t)a->H * a->W * a->Cin < img &&          // two images within the 2 GiB range of the input descriptor
           (size_t)4 * a->H * a->W * a->Cout < img &&     // ... and of the output descriptor
           (size_t)25 * a->Cin * a->Cout < ((size_t)1 << 29);
}
bool femasr_conv_wino_up2_shape_ok(const femasr_conv_args *a) { return femasr_conv_wino_up2_shape_ok_lim(a, FEMASR_WINO_LOG2_TOTAL, FEMASR_WINO_LOG2_IMAGE); }
const char *femasr_conv_wino_up2_variant_name() { return "conv3x3_wino_up2<2x16x16px x64,waves=8>"; }

int femasr_conv_wino_up2_launch(hipStream_t s, const femasr_conv_args *a, double *flops_out)
{
    FEMASR_REQUIRE(a && a->in && a->w_wino && a->bias && a->out && femasr_conv_wino_up2_shape_ok(a), "conv_wino_up2: bad arguments / shape");
    FEMASR_REQUIRE(a->Ho == 2 * a->H && a->Wo == 2 * a->W, "conv_wino_up2: Ho/Wo mismatch");
    FEMASR_REQUIRE(!a->gn_part || femasr_gn_fusable(a->Cout), "conv_wino_up2: gn_part needs 32 | Cout and Cout/32 a power of two <= 32");
    WinoUpParams p{};
    p.in = a->in; p.in2 = a->in_add; p.u = (const float *)a->w_wino; p.bias = a->bias;
    p.res1 = a->res1; p.res2 = a->res2; p.out = a->out; p.gn_part = a->gn_part;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.Ho = a->Ho; p.Wo = a->Wo;
    p.sbX = (p.Wo + 15) / 16;
    p.sbY = (p.Ho + 15) / 16;
    p.nsb = a->B * p.sbX * p.sbY;
    p.MB = (p.nsb + 1) / 2;
    p.NB = a->Cout / 64;
    p.nsteps = a->Cin / 8;
    p.NT32 = a->Cout / 32;
    FEMASR_REQUIRE(a->res1 || !a->res2, "conv_wino_up2: res2 without res1");
    const int nres = (a->res1 ? (a->res2 ? 2 : 1) : 0) + (a->in_add ? 3 : 0);      // (index into the instantiation table)
    const size_t lds = wino_up_lds_bytes();
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((__atomic_load_n(&g_attr_devs[nres], __ATOMIC_ACQUIRE) >> dev) & 1ull)) {
        FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)g_wu_kern[nres], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) __atomic_fetch_or(&g_attr_devs[nres], 1ull << dev, __ATOMIC_RELEASE);
    }
    hipLaunchKernelGGL(g_wu_kern[nres], dim3((unsigned)(p.MB * p.NB)), dim3(WU_NT), lds, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    // ALGORITHMIC flops (the definition's 9 taps per output pixel, like every other conv launcher); the kernel issues 25/144 of them
    if (flops_out) *flops_out = 2.0 * (double)a->B * a->Ho * a->Wo * 9.0 * (double)a->Cin * (double)a->Cout;
    return FEMASR_OK;
}

extern "C" {

#ifdef FEMASR_WINO_TT
int femasr_debug_wino_up2_ttbuf(unsigned long long *dev_buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wu_ttbuf), &dev_buf, sizeof(dev_buf));
}
#endif

size_t femasr_wino_up2_weight_floats(int O, int I) { return (I % 32) == 0 ? (size_t)(I / 8) * 25 * ((O + 31) / 32) * 256 : 0; }

int femasr_repack_oihw_wino_up2(void *stream, const float *in, int O, int I, float *out)
{
    FEMASR_REQUIRE(in && out && O > 0 && I > 0 && (I % 32) == 0, "repack_wino_up2: needs a 3x3 OIHW weight with I %% 32 == 0");
    const size_t total = femasr_wino_up2_weight_floats(O, I);
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(repack_wino_up2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, O, I, out, total);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

}  // extern "C"
