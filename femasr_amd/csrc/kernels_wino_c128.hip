// kernels_wino_c128.hip — the Winograd F(4x4, 3x3) conv of kernels_wino.hip for layers with Cout % 128 == 0: block = ONE 16x16-pixel
// sub-block x 128 output channels, K walked in steps of 16 input channels (round 4).
//
// Why a second block shape.  In kernels_wino.hip the work beside the MFMAs - staging (GroupNorm + SiLU), the input transform, the patch
// loads - is done once per (Winograd tile, input channel) of the block and is paid again by every 64-channel column block of the layer; the
// fp32 MFMA pipe shares the vector lanes with it (time = MFMA + VALU + exposed latency, tools/ubench/coexec.hip).  With the accumulator file
// fixed (36 components x 2048 (tile, channel) outputs = 144 registers per lane) the ratio is set by the output channels per block alone:
// 16 tiles x 128 channels instead of 32 x 64 halves the staging / transform work per MFMA (and reads the input once instead of twice for
// Cout = 128); the price is twice the weight-fragment traffic from L2 per MFMA (16 tiles share a fragment instead of 32).
//
// What changes against kernels_wino.hip (everything else - transforms, staging, epilogue arithmetic, every addition order - is the same code):
//   * a step = 16 input channels of the one sub-block.  The staging / transform index space is unchanged: what was "sub-block z" of the pair
//     is now "channel half z" (channels 8z .. 8z+7 of the step) of the same pixels: patch buffer [z][18x18][8 + 2], V[component][z][..].
//   * M phase: v_mfma_f32_16x16x1_f32 (4 blocks): rows = the 16 Winograd tiles, the 4 blocks = 4 x 16 output channels, ONE input channel per
//     instruction - the chain over the input channels is the sequence of instructions (no in-instruction k order to rely on), 32 cycles each,
//     the same 256 flop / cycle / CU as the 32x32x2 form.  Wave w owns 9 (component, 64-channel half) pairs as before: 16 MFMAs per pair and
//     step; the two halves of a component share the V fragment (one ds_read_b128 = 4 channels = 8 MFMAs).
//   * weight fragments: [step][component][64-channel half][channel quad][lane = output channel][4] - a 1-KiB contiguous wave load per 4
//     MFMAs, streamed from L2 through an 8-deep register ring that stays live across the T phase (8 loads = 1024 MFMA cycles ahead).  The
//     patch request of step s+3 goes out at the END of the M phase, behind every ring request of the step (loads return in order).
//   * epilogue: a round = one 64-channel half; thread = (tile pair = wave, channel = lane).  The GroupNorm partial moments keep the
//     order of orc_gn_wino_partial: fp32 per (tile, channel), fp64: tile pair, channels of the group (xor butterfly), the two pairs of the
//     tile row (now two waves: through LDS), the four tile rows in order.
// Bit-identical to kernels_wino.hip and to oracle/femasr_oracle.c orc_conv3x3_winograd: every output is the same sequence of operations.
#define FEMASR_WTT_BUF g_wc_ttbuf
#include "wino_common.h"

#ifndef FEMASR_WC_HOIST
#define FEMASR_WC_HOIST 0
#endif

namespace {

// V buffer of a step: [component 36][channel half z][channel quad kh][tile ^ 8 kh][4 channels] (floats): the fragment of a lane = tile
// lane % 16 is one ds_read_b128, 16 lanes read 256 contiguous bytes; the xor keeps the transform's stores of channels c and c + 4 of a tile
// in different banks.
__device__ __forceinline__ int vc_off(int z, int kh, int tile) { return z * 128 + kh * 64 + ((tile ^ (8 * kh)) << 2); }

template <int PRO, bool FAST, int NRES>
__global__ __launch_bounds__(W4_NT, 2) void conv3x3_wino4c_kernel(const WinoParams p)
{
    constexpr bool HAS1 = NRES >= 1, HAS2 = NRES >= 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                        // [2][W4_PSZ]: [channel half][18x18][W4_PS]
    float *Vs = smem + 2 * W4_PSZ;           // [2][W4_VSZ]
    float *ABs = smem + W4_MAIN;             // [a | b][Cin]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    WTT_INIT
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB, mb = L / p.NB;
    const int n0 = nb * 128;

    // ---- the sub-block (uniform)
    const int per = p.sbY * p.sbX;
    const int sn = mb / per, sbi = mb - sn * per;
    const int sy0 = 16 * (sbi / p.sbX), sx0 = 16 * (sbi - (sbi / p.sbX) * p.sbX);

    // ---- staging units: float4 = (channel half z, pixel of the 18x18 patch, channel quad t&1); unit u of the 2 x 648: u = t, t + 512, and 34
    // lanes per wave of a third round; LDS slot (u >> 1) * PS + 4 (u & 1) floats (the second half's patch lies right behind the first one's)
    const int quad = t & 1;
    unsigned goff[3];
    unsigned pmask = 0, rmask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = i < 2 ? t + 512 * i : 1024 + wave * 34 + lane;
        const bool real = i < 2 || lane < 34;
        const int z = u >= W4_UNITS_SB ? 1 : 0;
        const int pix = (u - W4_UNITS_SB * z) >> 1;
        const int py = pix / W4_PW, px = pix - py * W4_PW;
        const int y = sy0 - 1 + py, x = sx0 - 1 + px;
        const bool ok = real && y >= 0 && y < p.H && x >= 0 && x < p.W;
        goff[i] = ok ? (unsigned)(((size_t)y * p.W + x) * p.Cin + 8 * z + 4 * quad) * 4u : 0u;
        pmask |= (ok ? 1u : 0u) << i;
        rmask |= (real ? 1u : 0u) << i;
    }
    const int slot0 = (t >> 1) * W4_PS + 4 * quad;                       // unit 0; unit 1 = + 256 PS; unit 2 = + (512 - 15 wave) PS
    const int slot2d = (512 - 15 * wave) * W4_PS;                        // (uniform)
    auto unit_slot = [&](int i) -> int { return i == 0 ? slot0 : (i == 1 ? slot0 + 256 * W4_PS : slot0 + slot2d); };
    const int zo1 = t >= W4_UNITS_SB - 512 ? 8 : 0;                      // channel half of unit 1 (unit 0: first, unit 2: second)
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void *)(p.in + (size_t)sn * p.H * p.W * p.Cin), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, 0x7fffffff, 0x00020000);
    float4 rp[3], rq[3];
    auto load_patch_to = [&](float4 (&rr)[3], int s) {       // unconditional (steps past the end re-read the last one): the wait counters stay static
        const int sc = s < p.nsteps ? s : p.nsteps - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (FEMASR_WINO_ABL & 1) { rr[i] = make_float4(0.1f * sc, 0.2f, 0.3f, 0.4f); continue; }
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, goff[i], sc * 64, W_NT_IN);
            rr[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto store_patch_from = [&](const float4 (&rr)[3], int s, int buf) {
        float *Pb = Ps + buf * W4_PSZ;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (FEMASR_WINO_ABL & 64) { asm volatile("" :: "v"(rr[i].x), "v"(rr[i].y), "v"(rr[i].z), "v"(rr[i].w)); continue; }
            float4 v = rr[i];
            if (PRO == FEMASR_PRO_GN_SILU && !(FEMASR_WINO_ABL & 32)) {
                const float *ab = ABs + (i == 0 ? 0 : (i == 1 ? zo1 : 8)) + s * 16 + 4 * quad;
                const float4 ga = ld4(ab), gb = ld4(ab + p.Cin);
                if (FAST) {      // x * rcp(1 + exp2(-x log2 e)), two-wide wherever the instruction set is
                    const tf2 x01 = __builtin_elementwise_fma(tf2{v.x, v.y}, tf2{ga.x, ga.y}, tf2{gb.x, gb.y});
                    const tf2 x23 = __builtin_elementwise_fma(tf2{v.z, v.w}, tf2{ga.z, ga.w}, tf2{gb.z, gb.w});
                    const tf2 nl2e = {-1.44269504088896341f, -1.44269504088896341f}, one = {1.0f, 1.0f};
                    const tf2 e01 = x01 * nl2e, e23 = x23 * nl2e;
                    const tf2 d01 = tf2{__builtin_amdgcn_exp2f(e01[0]), __builtin_amdgcn_exp2f(e01[1])} + one;
                    const tf2 d23 = tf2{__builtin_amdgcn_exp2f(e23[0]), __builtin_amdgcn_exp2f(e23[1])} + one;
                    const tf2 y01 = x01 * tf2{__builtin_amdgcn_rcpf(d01[0]), __builtin_amdgcn_rcpf(d01[1])};
                    const tf2 y23 = x23 * tf2{__builtin_amdgcn_rcpf(d23[0]), __builtin_amdgcn_rcpf(d23[1])};
                    v = make_float4(y01[0], y01[1], y23[0], y23[1]);
                } else {
                    v.x = det_silu(__builtin_fmaf(v.x, ga.x, gb.x));
                    v.y = det_silu(__builtin_fmaf(v.y, ga.y, gb.y));
                    v.z = det_silu(__builtin_fmaf(v.z, ga.z, gb.z));
                    v.w = det_silu(__builtin_fmaf(v.w, ga.w, gb.w));
                }
            }
            if (pmask & (1u << i)) {                      // zero padding AFTER the activation: those slots keep their zeros
                float *dst = Pb + unit_slot(i);
                *reinterpret_cast<float2 *>(dst) = make_float2(v.x, v.y);
                *reinterpret_cast<float2 *>(dst + 2) = make_float2(v.z, v.w);
            }
        }
    };

    // ---- input transform item: (channel half, tile) tm, channel lane&7 of the half, rows 3*thalf .. 3*thalf+2 of B^T d (uniform per wave)
    const int thalf = wave & 1;
    const int tm = (wave >> 1) * 8 + (lane >> 3), tch = lane & 7;
    const int tsrc = (tm >> 4) * W4_PPIX * W4_PS + ((4 * ((tm & 15) >> 2) + thalf) * W4_PW + 4 * (tm & 3)) * W4_PS + tch;
    const int tdst = thalf * 18 * 256 + vc_off(tm >> 4, tch >> 2, tm & 15) + (tch & 3);
    tf2 td[5][3];
    auto transform_read = [&](int pbuf) {
        if (FEMASR_WINO_ABL & 4) return;
        const float *src = Ps + pbuf * W4_PSZ + tsrc;
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) td[a][b] = tf2{src[(a * W4_PW + 2 * b) * W4_PS], src[(a * W4_PW + 2 * b + 1) * W4_PS]};
    };
    auto fma2 = [](float c, tf2 x, tf2 y) -> tf2 { return __builtin_elementwise_fma(tf2{c, c}, x, y); };
    auto transform_write = [&](int vbuf) {
        if (FEMASR_WINO_ABL & 4) return;
        float *dst = Vs + vbuf * W4_VSZ + tdst;
        tf2 r[3][3];
        if (thalf == 0) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                r[0][b] = fma2(4.0f, td[0][b], fma2(-5.0f, td[2][b], td[4][b]));
                const tf2 a2 = fma2(-4.0f, td[2][b], td[4][b]), b2 = fma2(-4.0f, td[1][b], td[3][b]);
                r[1][b] = a2 + b2;
                r[2][b] = a2 - b2;
            }
        } else {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const tf2 c2 = td[3][b] - td[1][b], e2 = td[2][b] - td[0][b];
                r[0][b] = fma2(2.0f, e2, c2);
                r[1][b] = fma2(-2.0f, e2, c2);
                r[2][b] = fma2(4.0f, td[0][b], fma2(-5.0f, td[2][b], td[4][b]));
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float v0, v1, v2, v3, v4, v5;
            bt_lo(r[i][0][0], r[i][0][1], r[i][1][0], r[i][1][1], r[i][2][0], v0, v1, v2);
            bt_hi(r[i][0][1], r[i][1][0], r[i][1][1], r[i][2][0], r[i][2][1], v3, v4, v5);
            float *o = dst + i * 6 * 256;
            o[0 * 256] = v0;
            o[1 * 256] = v1;
            o[2 * 256] = v2;
            o[3 * 256] = v3;
            o[4 * 256] = v4;
            o[5 * 256] = v5;
        }
    };

    // ---- M phase: pair q < 8 = (component 4 wave + q/2, channel half q&1); pair 8 = (component 32 + wave/2, channel half wave&1).
    // acc[q][4 b + r] of lane l = M[tile 4 (l / 16) + r][channel 64 half + 16 b + l % 16]
    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned lw = (unsigned)lane * 16u;
    const int al0 = vc_off(0, 0, lane & 15), al1 = vc_off(0, 1, lane & 15);
    const int NH = p.NT32 >> 1;                       // 64-channel halves of the layer
    auto pcomp = [&](int q) -> int { return q < 8 ? 4 * wave + (q >> 1) : 32 + (wave >> 1); };
    auto pntl = [&](int q) -> int { return q < 8 ? (q & 1) : (wave & 1); };
    // weight quad g of step s, in the order the M phase consumes them: g < 32: component pair g / 8, channel quad (g / 2) % 4, half g % 2;
    // g = 32 .. 35: pair 8, channel quad g - 32; g >= 36 rolls into the next step (unconditional: past the end it re-reads the last step)
    auto ldUq = [&](int s, int g) -> f32x4_t {
        if (g >= 36) { g -= 36; s += 1; }
        const int sc = s < p.nsteps ? s : p.nsteps - 1;
        const int q = g < 32 ? (((g >> 3) << 1) | (g & 1)) : 8, j = g < 32 ? ((g >> 1) & 3) : g - 32;
        if (FEMASR_WINO_ABL & 2) { f32x4_t c = {0.5f + sc, 0.25f, 0.125f, 1.0f}; return c; }
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, lw, (((((sc * 36 + pcomp(q)) * NH + 2 * nb + pntl(q)) << 2) + j) << 10), 0);
        return __builtin_bit_cast(f32x4_t, v);
    };
    f32x4_t ring[8], an;
    // V fragment of (pair-group cg: 0-3 = the wave's components 4 wave + cg, 4 = pair 8's component; channel quad j), one group ahead
    auto ldA = [&](const float *Vb, int cgi, int j) -> f32x4_t {
        const int comp = cgi < 4 ? pcomp(2 * cgi) : pcomp(8);
        return *reinterpret_cast<const f32x4_t *>(Vb + comp * 256 + (j >> 1) * 128 + ((j & 1) ? al1 : al0));
    };
    // quads 4 .. 7 of step s: requested at the END of the T phase of step s-1 (in flight across the barrier; quads 0 .. 3 cover their latency) -
    // across the T phase only half of the ring is live, which is what keeps the staging / transform registers out of scratch
    auto issue_early = [&](int s, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
#pragma unroll
        for (int g = 4; g < 8; ++g) ring[(g + 4 * PAR) & 7] = ldUq(s, g);
    };
    auto mphase = [&](int s, auto par_c) {      // par_c = s & 1 at compile time: the ring slot of quad g is (g + 4 par) % 8 (36 quads per step)
        constexpr int PAR = decltype(par_c)::value;
        const float *Vb = Vs + PAR * W4_VSZ;
        an = ldA(Vb, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g0 = 8 * cp + 2 * j, s0 = (g0 + 4 * PAR) & 7, s1 = (g0 + 1 + 4 * PAR) & 7;
                const f32x4_t a = an, b0 = ring[s0], b1 = ring[s1];
                an = j < 3 ? ldA(Vb, cp, j + 1) : ldA(Vb, cp + 1, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (FEMASR_WINO_ABL & 8) { asm volatile("" :: "v"(a[e]), "v"(b0[e]), "v"(b1[e])); continue; }
                    acc[2 * cp] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[e], b0[e], acc[2 * cp], 0, 0, 0);
                    acc[2 * cp + 1] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[e], b1[e], acc[2 * cp + 1], 0, 0, 0);
                }
                ring[s0] = ldUq(s, g0 + 8);
                ring[s1] = ldUq(s, g0 + 9);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = 32 + j, sl = (g + 4 * PAR) & 7;
            const f32x4_t a = an, b = ring[sl];
            if (j < 3) an = ldA(Vb, 4, j + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (FEMASR_WINO_ABL & 8) { asm volatile("" :: "v"(a[e]), "v"(b[e])); continue; }
                acc[8] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[e], b[e], acc[8], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);       // (this slot's refill - quad 4 + j of the next step - goes out at the end of the T phase: issue_early)
        }
        // the patch of step s+3 (staged in the NEXT step's T phase, from the other register set): behind every weight request of the step
        if (PAR) load_patch_to(rp, s + 3); else load_patch_to(rq, s + 3);
    };

    // ---- prologue: every global request first, then the LDS work that needs none of them
    load_patch_to(rp, 0);
    load_patch_to(rq, 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) ring[g] = ldUq(0, g);
    if (PRO == FEMASR_PRO_GN_SILU) {     // GN table [a | b][Cin]
        for (int i = t; i < 2 * p.Cin; i += W4_NT) {
            const int kind = i >= p.Cin ? 1 : 0, c = i - kind * p.Cin;
            ABs[i] = (kind ? p.pro_b : p.pro_a)[(size_t)sn * p.Cin + c];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {        // the slots the main loop never writes hold the zero padding in both buffers
        if (((rmask & ~pmask) >> i) & 1u) {
            float *dst = Ps + unit_slot(i);
            *reinterpret_cast<float2 *>(dst) = make_float2(0.f, 0.f);
            *reinterpret_cast<float2 *>(dst + 2) = make_float2(0.f, 0.f);
            *reinterpret_cast<float2 *>(dst + W4_PSZ) = make_float2(0.f, 0.f);
            *reinterpret_cast<float2 *>(dst + W4_PSZ + 2) = make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    store_patch_from(rp, 0, 0);
    load_patch_to(rp, 2);
    if (1 < p.nsteps) store_patch_from(rq, 1, 1);
    __syncthreads();
    transform_read(0);
    transform_write(0);
    issue_early(0, std::integral_constant<int, 0>{});
    __syncthreads();
    WTT(0)

    // ---- main loop: every wave M(s), then T(s) (see kernels_wino.hip), one barrier per step
    constexpr bool HOIST = FEMASR_WC_HOIST != 0;
    auto step = [&](int s, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        mphase(s, par_c);
        if (HOIST) transform_read(PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (s < 24) WTT(64 + 2 * s)
        auto stage = [&]() {                          // the patch of step s+2 -> buffer PAR, from the set of this parity (requested a step ago)
            if (s + 2 < p.nsteps) { if (PAR) store_patch_from(rq, s + 2, PAR); else store_patch_from(rp, s + 2, PAR); }
        };
        if (HOIST) {
            transform_write(PAR ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            stage();
        } else {
            stage();
            __builtin_amdgcn_sched_barrier(0);
            transform_read(PAR ^ 1);
            transform_write(PAR ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_early(s + 1, std::integral_constant<int, PAR ^ 1>{});
        if (s < 24) WTT(65 + 2 * s)
        __syncthreads();
        if (s < 40) WTT(16 + s)
    };
    for (int s = 0; s < p.nsteps; s += 2) {           // (nsteps = Cin / 16 is even)
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
    }
    WTT(1)

    // ---------------------------------------------------------------------------------------------------------------
    // epilogue, one 64-channel half (round) at a time: accumulators -> Mx[component][tile pair 2 ty + tx/2][channel position][2]
    // (registers 4b + rr, 4b + rr + 1 are two horizontally adjacent tiles: one ds_write_b64; channel 16 b + c of tile row ty sits at
    // position 16 (b ^ ty) + c so that the four 16-lane groups of a store land in different bank windows).  A thread = (tile pair = wave,
    // channel = lane) handles both tiles at once; everything else as kernels_wino.hip.
    float *Mx = smem;
    double *red = reinterpret_cast<double *>(smem + W4_MX);
    const bool gnp = p.gn_part != nullptr;
    const int cg = p.Cout >> 5;                      // channels per GroupNorm group (4 .. 32: Cout % 128 == 0, Cout <= 1024)
    const int gph = 64 / cg;                         // groups per 64-channel half (<= 16)
    const int ety = wave >> 1, etx = 2 * (wave & 1); // tile row, left tile of the pair (uniform)
    unsigned vmask[2] = {0xffffu, 0xffffu}, ooff;
    const bool full = sy0 + 16 <= p.H && sx0 + 16 <= p.W;      // (uniform)
    {
        const int oy = sy0 + 4 * ety, ox = sx0 + 4 * etx;
        if (!full) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                unsigned m = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) m |= (oy + (k >> 2) < p.H && ox + 4 * e + (k & 3) < p.W ? 1u : 0u) << k;
                vmask[e] = m;
            }
        }
        ooff = (unsigned)(((size_t)oy * p.W + ox) * p.Cout + n0 + lane) * 4u;      // (< 2^29: one image of < 2^27 elements)
    }
    const size_t img0 = (size_t)sn * p.H * p.W * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out + img0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc((void *)((HAS1 ? p.res1 : p.out) + img0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r2 = __builtin_amdgcn_make_buffer_rsrc((void *)((HAS2 ? p.res2 : p.out) + img0), 0, 0x7fffffff, 0x00020000);
    int cs_u = 0, rs_u = 0;      // set (opaque) at the top of every round
    auto soff = [&](int k, int e, int r) -> int { return (k >> 2) * rs_u + (4 * e + (k & 3)) * cs_u + 256 * r; };
    auto oob = [&](int e, int k) -> unsigned { return ~(unsigned)((int)(vmask[e] << (31 - k)) >> 31); };
    auto voff = [&](auto fullc, int e, int k) -> unsigned {
        if (decltype(fullc)::value) return ooff;
        return ooff | oob(e, k);
    };
    auto fetch = [&](auto fullc, const __amdgpu_buffer_rsrc_t rs, int r, tf2 (&dst)[16], int k0, int k1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < k0 || k >= k1) continue;
            dst[k][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff(fullc, 0, k), soff(k, 0, r), W_NT_IO));
            dst[k][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff(fullc, 1, k), soff(k, 1, r), W_NT_IO));
        }
    };
    const int ridx = wave * 64 + (lane ^ (16 * ety));      // this thread's (tile pair, channel position) in a component's 4-KiB plane
    auto round = [&](auto fullc, int r) {
        constexpr bool FULL = decltype(fullc)::value;
        tf2 r1[16], r2[16];
        if (!FULL) asm volatile("" : "+v"(vmask[0]), "+v"(vmask[1]));
        cs_u = __builtin_amdgcn_readfirstlane(p.Cout * 4);
        rs_u = __builtin_amdgcn_readfirstlane(p.W * p.Cout * 4);
        asm volatile("" : "+s"(cs_u), "+s"(rs_u));
        if (FULL) {          // (a border block fetches row by row below: kernels_wino.hip)
            if (HAS1) fetch(fullc, rs_r1, r, r1, 0, HAS2 ? 8 : 16);
            if (HAS2) fetch(fullc, rs_r2, r, r2, 0, 8);
        }
        const float bv = p.bias[n0 + 64 * r + lane];
        __syncthreads();
        WTT(3 + 4 * r)
        if (!(FEMASR_WINO_ABL & 16)) {
            int so1 = (16 * 512 + ridx) * 8, so2 = (32 * 512 + ridx) * 8;
            asm volatile("" : "+v"(so1), "+v"(so2));
            const tf2 *src0 = reinterpret_cast<const tf2 *>(Mx) + ridx;
            const tf2 *src1 = reinterpret_cast<const tf2 *>(reinterpret_cast<const char *>(Mx) + so1), *src2 = reinterpret_cast<const tf2 *>(reinterpret_cast<const char *>(Mx) + so2);
            auto mx = [&](int c) -> tf2 { return c < 16 ? src0[c * 512] : (c < 32 ? src1[(c - 16) * 512] : src2[(c - 32) * 512]); };
            tf2 tt[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                at6(mx(0 * 6 + j), mx(1 * 6 + j), mx(2 * 6 + j), mx(3 * 6 + j), mx(4 * 6 + j), mx(5 * 6 + j), tt[0][j], tt[1][j], tt[2][j], tt[3][j]);
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
                at6(tt[a][0], tt[a][1], tt[a][2], tt[a][3], tt[a][4], tt[a][5], y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = 4 * a + b;
                    tf2 v = y[b] + bv2;
                    if (HAS1) v = v + r1[k];
                    if (HAS2) v = v + r2[k];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), rs_out, voff(fullc, 0, k), soff(k, 0, r), W_NT_IO);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1]), rs_out, voff(fullc, 1, k), soff(k, 1, r), W_NT_IO);
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
            if (gnp) {      // fp64 from here: the two tiles of the pair, then the channels of the group
                double gs = (double)s2[0] + (double)s2[1], gss = (double)ss2[0] + (double)ss2[1];
                for (int d = 1; d < cg; d <<= 1) {
                    gs = gs + __shfl_xor(gs, d, 64);
                    gss = gss + __shfl_xor(gss, d, 64);
                }
                if ((lane & (cg - 1)) == 0) {
                    double *dst = red + ((size_t)wave * 16 + lane / cg) * 2;
                    dst[0] = gs;
                    dst[1] = gss;
                }
            }
        }
        WTT(4 + 4 * r)
    };
    // accumulator stores: four per-lane byte offsets (one per 16-channel block b, the xor is per lane), pinned as integers; component and
    // the right pair of the tile row are immediates
    int wo[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wo[b] = (4 * wave) * 4096 + ((2 * (lane >> 4)) * 64 + 16 * (b ^ (lane >> 4)) + (lane & 15)) * 8;
    asm volatile("" : "+v"(wo[0]), "+v"(wo[1]), "+v"(wo[2]), "+v"(wo[3]));
    const int wd8 = (32 + (wave >> 1) - 4 * wave) * 4096;           // pair 8's component against the wave's first (uniform)
    auto write_acc = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (pntl(q) == R) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    char *dst = reinterpret_cast<char *>(Mx) + wo[b] + (q < 8 ? (q >> 1) * 4096 : wd8);
                    *reinterpret_cast<tf2 *>(dst) = tf2{acc[q][4 * b], acc[q][4 * b + 1]};
                    *reinterpret_cast<tf2 *>(dst + 512) = tf2{acc[q][4 * b + 2], acc[q][4 * b + 3]};
                }
            }
        }
    };
    write_acc(std::integral_constant<int, 0>{});
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        if (r == 1) write_acc(std::integral_constant<int, 1>{});
        WTT(2 + 4 * r)
        if (full) round(std::true_type{}, r); else round(std::false_type{}, r);
        __syncthreads();
        WTT(5 + 4 * r)
        if (gnp && t < gph) {       // the two pairs of a tile row, then the four tile rows in order
            double S = red[((size_t)0 * 16 + t) * 2] + red[((size_t)1 * 16 + t) * 2];
            double SS = red[((size_t)0 * 16 + t) * 2 + 1] + red[((size_t)1 * 16 + t) * 2 + 1];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                S = S + (red[((size_t)(2 * w) * 16 + t) * 2] + red[((size_t)(2 * w + 1) * 16 + t) * 2]);
                SS = SS + (red[((size_t)(2 * w) * 16 + t) * 2 + 1] + red[((size_t)(2 * w + 1) * 16 + t) * 2 + 1]);
            }
            const int g = (n0 + 64 * r) / cg + t;
            double *dst = p.gn_part + (((size_t)sn * p.sbY * p.sbX + sbi) * 32 + g) * 2;
            dst[0] = S;
            dst[1] = SS;
        }
    }
    WTT_END
}

// 3x3 OIHW -> out[step = ci/16][component 6 i + j][64-channel half][channel quad kq][lane][e]: o = 64 half + lane, ci = 16 step + 4 kq + e
// (the B operands of 4 consecutive MFMAs: one 16-byte load per lane, 1 KiB per wave); G g G^T as in kernels_wino.hip (wino_g6).
__global__ void repack_wino_c128_kernel(const float *__restrict__ in, int O, int I, float *__restrict__ out, size_t total)
{
    const int NH = O / 64;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), kq = (int)((idx >> 8) & 3);
        size_t rest = idx >> 10;
        const int half = (int)(rest % NH);
        rest /= NH;
        const int comp = (int)(rest % 36), step = (int)(rest / 36);
        const int ci = 16 * step + 4 * kq + e, o = 64 * half + lane;
        const int i = comp / 6, j = comp - 6 * i;
        const float *gw = in + ((size_t)o * I + ci) * 9;
        float ur[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) ur[x] = wino_g6(i, gw[x], gw[3 + x], gw[6 + x]);
        out[idx] = wino_g6(j, ur[0], ur[1], ur[2]);
    }
}

struct WCVariant {
    const char *name;
    void (*kern)(const WinoParams);
    unsigned long long attr_devs;
    size_t attr_lds;
};
#define FEMASR_WINOC(PRO, FAST, NRES) { "conv3x3_wino4<16x16px x128," #PRO "," #FAST ",res=" #NRES ",waves=8>", conv3x3_wino4c_kernel<PRO, FAST, NRES>, 0ull, 0 }
#define FEMASR_WINOC3(PRO, FAST) FEMASR_WINOC(PRO, FAST, 0), FEMASR_WINOC(PRO, FAST, 1), FEMASR_WINOC(PRO, FAST, 2)
WCVariant g_wc[] = {                              // index = 3 * (prologue form) + residual operands, as kernels_wino.hip
    FEMASR_WINOC3(FEMASR_PRO_NONE, false),
    FEMASR_WINOC3(FEMASR_PRO_GN_SILU, false),
    FEMASR_WINOC3(FEMASR_PRO_GN_SILU, true),
};
constexpr int kNumWC = sizeof(g_wc) / sizeof(g_wc[0]);

std::atomic<int> g_form{-1};       // -1: ask the environment (FEMASR_WINO_C128, default 0), 0: the x64 form everywhere, 1: x128 where the shape allows

int form_enabled()
{
    int v = g_form.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("FEMASR_WINO_C128");
        v = (e && atoi(e) != 0) ? 1 : 0;
        g_form.store(v, std::memory_order_relaxed);
    }
    return v;
}

}  // namespace

// the layers that take this block shape: decided by (Cin, Cout) alone, so that the packed weights of a layer have one layout
bool femasr_wino_c128_shape(int Cin, int Cout) { return form_enabled() && (Cout % 128) == 0 && (Cin % 32) == 0; }
void femasr_wino_c128_set_form(int on) { g_form.store(on < 0 ? -1 : (on ? 1 : 0), std::memory_order_relaxed); }
int femasr_conv_wino_c128_variant_count() { return kNumWC; }
const char *femasr_conv_wino_c128_variant_name(int v) { return v >= 0 && v < kNumWC ? g_wc[v].name : "?"; }

int femasr_repack_oihw_wino_c128(hipStream_t s, const float *in, int O, int I, float *out, size_t total)
{
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(repack_wino_c128_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, O, I, out, total);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

// (arguments validated by femasr_conv_wino_launch, which routes here when femasr_wino_c128_shape(Cin, Cout))
int femasr_conv_wino_c128_launch(hipStream_t s, const femasr_conv_args *a, int *variant_out)
{
    const bool gn = a->prologue == FEMASR_PRO_GN_SILU;
    WinoParams p{};
    p.in = a->in; p.u = (const float *)a->w_wino; p.bias = a->bias; p.pro_a = a->pro_a; p.pro_b = a->pro_b;
    p.res1 = a->res1; p.res2 = a->res2; p.out = a->out; p.gn_part = a->gn_part;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
    p.sbX = (a->W + 15) / 16;
    p.sbY = (a->H + 15) / 16;
    p.nsb = a->B * p.sbX * p.sbY;
    p.MB = p.nsb;
    p.NB = a->Cout / 128;
    p.nsteps = a->Cin / 16;
    p.NT32 = a->Cout / 32;
    const int vi = 3 * (gn ? (a->fast_act ? 2 : 1) : 0) + (a->res1 ? (a->res2 ? 2 : 1) : 0);
    WCVariant &v = g_wc[vi];
    const size_t main_f = (size_t)W4_MAIN + (gn ? 2 * (size_t)a->Cin : 0), epi_f = (size_t)W4_MX + W4_RED;
    const size_t lds = (main_f > epi_f ? main_f : epi_f) * sizeof(float);
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !((v.attr_devs >> dev) & 1ull) || v.attr_lds < lds) {
            const size_t want = v.attr_lds > lds ? v.attr_lds : lds;
            FEMASR_CHECK_HIP(hipFuncSetAttribute((const void *)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
            if (v.attr_lds < want) { v.attr_lds = want; v.attr_devs = 0ull; }
            if (dev >= 0 && dev < 64) v.attr_devs |= 1ull << dev;
        }
    }
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(p.MB * p.NB)), dim3(W4_NT), lds, s, p);
    FEMASR_CHECK_HIP(hipGetLastError());
    if (variant_out) *variant_out = vi;
    return FEMASR_OK;
}

#ifdef FEMASR_WINO_TT
extern "C" int femasr_debug_wino_c128_ttbuf(unsigned long long *dev_buf)      // [blocks][2][128] on the device, zero-filled by the caller
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wc_ttbuf), &dev_buf, sizeof(dev_buf));
}
#endif
