// kernels_wino.hip — 3x3 stride-1 pad-1 convs in the Winograd F(4x4, 3x3) form on the fp32 matrix cores.
//
// Used for the convs BEHIND the codebook lookup of single-codebook networks (after_quant, decoder ResBlocks, the LQ
// encoder's up-blocks that only make skip features; fema_utils.py:65-84, femasr_arch.py:196-207,298): they cannot move a VQ
// index, and the form needs 36 multiplies per 4x4 outputs where the direct sweep needs 144 (4x fewer MFMAs; round 2 ran
// F(2x2,3x3): 2.25x).  All fp32; the order of every addition is the one of oracle/femasr_oracle.c orc_conv3x3_winograd, so
// the result is bit-identical to that restatement (measured against the reference goldens, whole network: <= 1.3e-5 max-abs on
// outputs of magnitude <= 2.7 and 1.6e-4 on the +-49 un-scaled case, where the direct form has 1.2e-5 / 1.5e-4:
// profiles/r03_parity_report.txt).
//
// What shapes the kernel (tools/ubench/coexec.hip, quantum.hip, round 3): v_mfma_f32_32x32x2_f32 runs on the vector lanes.
// A VALU instruction of ANY wave on a SIMD takes its cycles out of that SIMD's fp32 MFMA time (4 MFMA waves + 4 VALU waves
// take exactly the SUM of the two alone), and beside a wave that streams back-to-back MFMAs the partner's VALU work does not
// advance at all.  Only LDS traffic and global loads overlap with MFMAs.  So: time = MFMA + VALU + exposed latency -
// as few VALU instructions as possible, and every load issued a phase ahead of its use.
//
// Block = 8 waves (2 per SIMD), two 16x16-pixel sub-blocks (consecutive in (n, y, x) order; 2 x 16 Winograd tiles = ONE
// 32-row MFMA tile) x 64 output channels.  K is walked in steps of 8 input channels; per step
//   staging   the 2 x 18x18 halo patches of step s+2: global -> registers (under the MFMAs) -> GroupNorm-apply + SiLU ->
//             LDS (zeros outside the image, AFTER the activation)
//   T phase   thread = (Winograd tile, channel, half of the 6 transform rows): V = B^T d B for step s+1 into
//             V[36 components][32 tiles][8 channels] (channel order = the MFMA A fragment of 4 k-pairs: one ds_read_b128)
//   M phase   wave w owns 9 (component, 32-column tile) pairs - components 4w..4w+3 x both column tiles, and column tile w&1
//             of component 32 + w/2: M += V_k[32 x 8] . U_k[8 x 32], 4 MFMAs per pair,
//             U fragments stream from L2 through a register ring (1 KiB contiguous per wave load, issued one pair-group ahead)
//   one barrier per step; every wave runs M then T (VALU work cannot hide behind a partner's MFMAs, see the main loop).
// Epilogue: accumulators -> LDS per 32-column tile; thread = (Winograd tile, channel) applies A^T M A, bias, residuals,
// stores, and accumulates the fused GroupNorm partial moments (fp64) per 16x16 sub-block in the order of orc_gn_coeffs mode 2.
#define FEMASR_WTT_BUF g_wi_ttbuf
#include "wino_common.h"
#ifndef FEMASR_WINO_EPI_FENCE
#define FEMASR_WINO_EPI_FENCE 1
#endif

namespace {

// SiLU of the staged input.  FAST: x * rcp(1 + exp2(-x log2 e)) on the hardware transcendental units (v_exp_f32 / v_rcp_f32,
// 1 ulp each) - the default of the model: these convs sit behind the codebook lookup, the result stays within ~1e-6 relative of
// the exact form.  !FAST: the IEEE-exact polynomial + division of detmath.h, bit-identical to the oracle (decoder_math 'fp32_strict').
// (Round 5 built the M phase on the bf16 matrix pipe as a fourth template parameter - V split into three bf16 terms where its fragment is
// read, U pre-split at pack time, six v_mfma_f32_32x32x8_bf16_1k per (component, column tile) and step: fp32-grade, 4 % SLOWER per launch
// (profiles/r05_wino_bf16m_ab.txt; DESIGN.md 5 "Round 5" 6.) - and round 6 removed it from the tree: git history has it.)
template <int PRO, bool FAST, int NRES>
__global__ __launch_bounds__(W4_NT, 2) void conv3x3_wino4_kernel(const WinoParams p)
{
    constexpr bool HAS1 = NRES >= 1, HAS2 = NRES >= 2;      // residual operands of the epilogue (compile time: no selects per pixel)
        extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                        // [2][W4_PSZ]
    float *Vs = smem + 2 * W4_PSZ;           // [2][W4_VSZ]
    float *ABs = smem + W4_MAIN;             // [2 sub-blocks][a | b][Cin]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    WTT_INIT
    const int c31 = lane & 31, hh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, p.MB * p.NB);
    const int nb = L % p.NB, mb = L / p.NB;
    const int n0 = nb * 64;

    // ---- the two sub-blocks (uniform)
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

    // ---- staging units: float4 = (pixel of the 18x18 patch, channel quad t&1); unit u of the 2 x 648: u = t, t + 512, and 34 lanes
    // per wave of a third round.  The LDS slot of unit u is (u >> 1) * PS + 4 (u & 1) floats whatever its sub-block - the second
    // sub-block's patch lies right behind the first one's - so ONE per-lane offset serves the three rounds (+ a constant for the
    // second, + a wave-uniform term for the third).  A unit outside the image (zero padding, invalid sub-block) never changes: its
    // slot is zeroed once in both buffers and the main loop skips its store under the exec mask (no per-value selects).
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
        const int y = (z ? sy0[1] : sy0[0]) - 1 + py, x = (z ? sx0[1] : sx0[0]) - 1 + px;
        const bool ok = real && (z ? sval[1] : sval[0]) && y >= 0 && y < p.H && x >= 0 && x < p.W;
        // byte offset from the first sub-block's image (the second one lies in the same or the next image)
        goff[i] = ok ? (unsigned)((((size_t)(z ? sn[1] - sn[0] : 0) * p.H + y) * p.W + x) * p.Cin + 4 * quad) * 4u : 0u;
        pmask |= (ok ? 1u : 0u) << i;
        rmask |= (real ? 1u : 0u) << i;
    }
    const int slot0 = (t >> 1) * W4_PS + 4 * quad;                       // unit 0; unit 1 = + 256 PS; unit 2 = + (512 - 15 wave) PS
    const int slot2d = (512 - 15 * wave) * W4_PS;                        // (uniform)
    auto unit_slot = [&](int i) -> int { return i == 0 ? slot0 : (i == 1 ? slot0 + 256 * W4_PS : slot0 + slot2d); };
    constexpr int KINDS = 2;                                             // GN table rows per sub-block: a, b
    const int zo1 = t >= W4_UNITS_SB - 512 ? KINDS * p.Cin : 0;          // table offset of unit 1's sub-block (unit 0: first, unit 2: second)
    // buffer loads: descriptor + uniform byte offset in SGPRs, ONE 32-bit per-lane offset register per load (no 64-bit VALU adds)
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void *)(p.in + (size_t)sn[0] * p.H * p.W * p.Cin), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, 0x7fffffff, 0x00020000);
    float4 rp[3];
    auto load_patch_to = [&](float4 (&rr)[3], int s) {       // unconditional (steps past the end re-read the last one): the wait counters stay static
        const int sc = s < p.nsteps ? s : p.nsteps - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (FEMASR_WINO_ABL & 1) { rr[i] = make_float4(0.1f * sc, 0.2f, 0.3f, 0.4f); continue; }
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, goff[i], sc * 32, W_NT_IN);
            rr[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto load_patch = [&](int s) { load_patch_to(rp, s); };
    auto store_patch_from = [&](const float4 (&rr)[3], int s, int buf) {
        float *Pb = Ps + buf * W4_PSZ;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (FEMASR_WINO_ABL & 64) { asm volatile("" :: "v"(rr[i].x), "v"(rr[i].y), "v"(rr[i].z), "v"(rr[i].w)); continue; }
            float4 v = rr[i];
            if (PRO == FEMASR_PRO_GN_SILU && !(FEMASR_WINO_ABL & 32)) {
                const float *ab = ABs + (i == 0 ? 0 : (i == 1 ? zo1 : KINDS * p.Cin)) + s * 8 + 4 * quad;
                const float4 ga = ld4(ab), gb = ld4(ab + p.Cin);
                if (FAST) {      // x * rcp(1 + exp2(-x log2 e)), two-wide wherever the instruction set is (v_pk_fma / v_pk_mul / v_pk_add)
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
    // Two register sets (FEMASR_WINO_DEEP): the patch requested in step s - at pair 5, behind the step's U requests, because loads return
    // in order - is the one staged in step s+1, a whole step later: its HBM latency is never waited for.  (With one set the request of
    // pair 5 was waited for in the same step's T phase; removing the patch loads altogether was worth 10 %, tools/build_debug.sh abl1.)
    // The set a step stages is a COMPILE-TIME choice (the main loop is unrolled by two): the wait-count bookkeeping is per register.
    constexpr bool DEEP = FEMASR_WINO_DEEP != 0;
    float4 rq[3];

    // ---- input transform item: tile tm, channel lane&7, rows 3*thalf .. 3*thalf+2 of B^T d (uniform per wave)
    const int thalf = wave & 1;
    const int tm = (wave >> 1) * 8 + (lane >> 3), tch = lane & 7;
    const int tsrc = (tm >> 4) * W4_PPIX * W4_PS + ((4 * ((tm & 15) >> 2) + thalf) * W4_PW + 4 * (tm & 3)) * W4_PS + tch;
    const int tdst = thalf * 18 * 256 + tm * 8 + (tch & 1) * 4 + (tch >> 1);
    // (first pass written on column PAIRS, explicitly two-wide: the patch reads come back as register pairs in exactly this order
    // (ds_read2_b32) and v_pk_fma / v_pk_add take them as they are; left to the vectoriser the same arithmetic cost 44 v_mov per step)
    tf2 td[5][3];                                          // the transform item's 5 x 6 patch values (rows 3*thalf.. of the 6x6 patch)
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
        if (thalf == 0) {        // bt_lo on rows 0..4, two columns at a time (component-wise the same IEEE sequence)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                r[0][b] = fma2(4.0f, td[0][b], fma2(-5.0f, td[2][b], td[4][b]));
                const tf2 a2 = fma2(-4.0f, td[2][b], td[4][b]), b2 = fma2(-4.0f, td[1][b], td[3][b]);
                r[1][b] = a2 + b2;
                r[2][b] = a2 - b2;
            }
        } else {                 // bt_hi on rows 1..5 (held as td[0..4])
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

    // ---- M phase: pair q < 8 = (component 4 wave + q/2, column tile q&1); pair 8 = (component 32 + wave/2, column tile wave&1)
    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned lw = (unsigned)lane * 16u;
    const int aoff = c31 * 8 + hh * 4;
    auto pcomp = [&](int q) -> int { return q < 8 ? 4 * wave + (q >> 1) : 32 + (wave >> 1); };
    auto pntl = [&](int q) -> int { return q < 8 ? (q & 1) : (wave & 1); };
    auto ldU = [&](int s, int q) -> u32x4_t {       // unconditional, like load_patch
        const int sc = s < p.nsteps ? s : p.nsteps - 1;
        if (FEMASR_WINO_ABL & 2) return u32x4_t{0x3f003f80u, 0x3e803e00u, 0x3f003f80u, 0x3e803e00u};
        const int pr = (sc * 36 + pcomp(q)) * p.NT32 + 2 * nb + pntl(q);
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, lw, pr << 10, 0);
    };
    // U fragments.  ring: pairs 0-2 of the NEXT step, loaded behind this step's pairs 0-2 (live across the T phase).
    // early: pairs 3-5, issued at the end of the T phase (the registers are free again): in flight across the barrier for
    // waves 0-3, whose M phase follows it.
    // late: pairs 6-8, issued at the start of the M phase (6 pairs = 1536 MFMA cycles ahead of their use).
    u32x4_t ring[3], early[3], late[3];
    auto issue_early = [&](int s) {
#pragma unroll
        for (int q = 3; q < 6; ++q) early[q - 3] = ldU(s, q);
    };
    auto mphase = [&](int s, auto par_c) {      // par_c = s & 1 at compile time
        constexpr int PAR = decltype(par_c)::value;
        const float *Vb = Vs + PAR * W4_VSZ + aoff;
#pragma unroll
        for (int q = 6; q < 9; ++q) late[q - 6] = ldU(s, q);
        f32x4_t an = *reinterpret_cast<const f32x4_t *>(Vb + pcomp(0) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const f32x4_t a = an;
            if ((q & 1) && q + 1 < 9) an = *reinterpret_cast<const f32x4_t *>(Vb + pcomp(q + 1) * 256);      // one A fragment per component
            const f32x4_t b = __builtin_bit_cast(f32x4_t, q < 3 ? ring[q] : (q < 6 ? early[q - 3] : late[q - 6]));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (FEMASR_WINO_ABL & 8) { asm volatile("" :: "v"(a[e]), "v"(b[e])); continue; }
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[q], 0, 0, 0);
            }
            if (q < 3) ring[q] = ldU(s + 1, q);
            if (q == 5) {      // behind the step's U requests (earlier it delays them: loads return in order)
                if (!DEEP) load_patch(s + 2);                     // staged in THIS step's T phase: 3 pairs + the transform ahead of its use
                else if (PAR) load_patch_to(rp, s + 3);           // staged in the NEXT step's T phase, from the other set
                else load_patch_to(rq, s + 3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: every global request first (patch of step 0 - and, variant bit 4, of step 1 in a second register set - the first U
    // fragments, the GN table), then the LDS work that needs none of them
    load_patch(0);
    load_patch_to(rq, 1);
#pragma unroll
    for (int q = 0; q < 3; ++q) ring[q] = ldU(0, q);
    if (PRO == FEMASR_PRO_GN_SILU) {     // GN table [sub-block][a | b][Cin]
        for (int i = t; i < 2 * KINDS * p.Cin; i += W4_NT) {
            const int z = i / (KINDS * p.Cin), r = i - z * KINDS * p.Cin, kind = r / p.Cin, c = r - kind * p.Cin;
            const float v = ((kind & 1) ? p.pro_b : p.pro_a)[(size_t)(z ? sn[1] : sn[0]) * p.Cin + c];
            ABs[i] = v;
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
    if (DEEP) load_patch_to(rp, 2);              // (set 0 is free again: the patch step 0 stages)
    if (1 < p.nsteps) store_patch_from(rq, 1, 1);
    __syncthreads();
    transform_read(0);
    transform_write(0);
    issue_early(0);
    __syncthreads();
    WTT(0)

    // ---- main loop.  Measured (tools/ubench/quantum.hip): while one wave of a SIMD streams back-to-back fp32 MFMAs, its
    // partner's VALU instructions do not issue at all (neither wave age nor s_setprio changes that) - the MFMA stream owns
    // the vector lanes.  VALU work therefore never hides behind another wave's MFMAs; time = MFMA + VALU + exposed latency.
    // A staggered order (waves 0-3 M then T, waves 4-7 T then M, with an s_sleep after every MFMA group so that the partner's
    // VALU advances in the gap) was built and measured 3-5 % SLOWER than this plain order: every wave M(s), then T(s).
    constexpr bool HOIST = FAST || PRO == FEMASR_PRO_NONE;       // the lean variants have the registers to read the next transform's
    auto step = [&](int s, auto par_c) {                         // patch values right behind the last MFMA issue
        constexpr int PAR = decltype(par_c)::value;
        mphase(s, par_c);
        if (HOIST) transform_read(PAR ^ 1);           // (unconditional: after the last step it transforms a stale patch into a dead buffer)
        __builtin_amdgcn_sched_barrier(0);
        if (s < 24) WTT(64 + 2 * s)
        auto stage = [&]() {                          // the patch of step s+2 -> buffer PAR (DEEP: from the set of this parity, requested a step ago)
            if (s + 2 < p.nsteps) { if (DEEP && PAR) store_patch_from(rq, s + 2, PAR); else store_patch_from(rp, s + 2, PAR); }
        };
        if (HOIST) {
            transform_write(PAR ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            stage();
        } else {
            stage();                                  // first: frees the staging registers ahead of the transform
            __builtin_amdgcn_sched_barrier(0);
            transform_read(PAR ^ 1);
            transform_write(PAR ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_early(s + 1);                                       // in flight across the barrier
        if (s < 24) WTT(65 + 2 * s)
        __syncthreads();
        if (s < 40) WTT(16 + s)
    };
    for (int s = 0; s < p.nsteps; s += 2) {           // (nsteps = Cin / 8 is a multiple of 4)
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
    }
    WTT(1)

    // ---------------------------------------------------------------------------------------------------------------
    // epilogue, one 32-column tile (round) at a time: accumulators -> Mx[component][tile pair][channel][2] (overlays the main-loop
    // buffers; rows e, e + 1 of an accumulator tile are two horizontally adjacent Winograd tiles and sit in an aligned register
    // pair: one ds_write_b64).  A thread = (tile pair pi = (sub-block, tile row, left / right half), channel c31) handles BOTH tiles
    // at once: one ds_read_b64 per component, and every operation of the output transform, the bias / residual adds and the moment
    // sums is a packed fp32 instruction over the pair (component-wise IEEE: the same values as two scalar sequences).  Outputs and
    // residuals go through buffer instructions: descriptors in SGPRs (base = the image of sub-block 0), ONE 32-bit byte offset per
    // lane, a uniform offset per pixel and tile of the pair; in a block whose 32 tiles lie wholly inside the image (FULL, the common
    // case) nothing is masked; otherwise a pixel outside gets an offset beyond num_records (stores dropped, loads return 0).
    // GroupNorm partial moments of what was stored (orc_gn_wino_partial): per (tile, channel) an fp32 sum and an fp32 fma chain of
    // squares over the 16 pixels (row-major, a pixel outside adds +0), then fp64: the two tiles of the pair, the channels of the
    // group (xor butterfly), the two pairs of the tile row, then the four tile rows (= waves) of the sub-block in order.
    float *Mx = smem;
    double *red = reinterpret_cast<double *>(smem + W4_MX);
    const bool gnp = p.gn_part != nullptr;
    const int cg = p.Cout >> 5;                      // channels per GroupNorm group (>= 2: Cout % 64 == 0)
    const int gpt = 32 / (cg < 32 ? cg : 32);        // groups per 32-channel tile (<= 16)
    const int pi = t >> 5;                           // tile pair: rows 2 pi, 2 pi + 1 of the accumulator tiles
    const int ez = wave >> 2, ety = wave & 3, etx = 2 * hh;      // sub-block (uniform), tile row (uniform), left tile of the pair
    unsigned vmask[2] = {0xffffu, 0xffffu}, ooff;
    bool full = true;
#pragma unroll
    for (int z = 0; z < 2; ++z) full = full && (z ? sval[1] : sval[0]) && (z ? sy0[1] : sy0[0]) + 16 <= p.H && (z ? sx0[1] : sx0[0]) + 16 <= p.W;      // (uniform)
    {
        const int oy = (ez ? sy0[1] : sy0[0]) + 4 * ety, ox = (ez ? sx0[1] : sx0[0]) + 4 * etx;
        if (!full) {         // per-pixel validity of the thread's two tiles (bit 4a + b): only a block that touches the image border needs it
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                unsigned m = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) m |= ((ez ? sval[1] : sval[0]) && oy + (k >> 2) < p.H && ox + 4 * e + (k & 3) < p.W ? 1u : 0u) << k;
                vmask[e] = m;
            }
        }
        ooff = (unsigned)((((ez ? sn[1] - sn[0] : 0) * p.H + oy) * p.W + ox) * p.Cout + n0 + c31) * 4u;      // (< 2^30: two images of < 2^27 elements)
    }
    const size_t img0 = (size_t)sn[0] * p.H * p.W * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out + img0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc((void *)((HAS1 ? p.res1 : p.out) + img0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r2 = __builtin_amdgcn_make_buffer_rsrc((void *)((HAS2 ? p.res2 : p.out) + img0), 0, 0x7fffffff, 0x00020000);
    // uniform byte offset of pixel k = 4a + b of tile e of the pair, round r: a * rs + (4e + b) * cs + 128 r.  The two strides are made
    // opaque once per round so that the 32 sums are formed where they are used (two scalar adds each) - hoisted out of the round loop
    // they do not fit the scalar registers and come back through v_readlane, which stalls every load / store on a scalar dependency
    int cs_u = 0, rs_u = 0;      // set (opaque) at the top of every round
    auto soff = [&](int k, int e, int r) -> int { return (k >> 2) * rs_u + (4 * e + (k & 3)) * cs_u + 128 * r; };
    // !FULL: all-ones where pixel k of tile e lies outside the image (two shifts on the lane's mask word; OR-ed into the offset,
    // AND-NOT-ed into the moment operand) - per-lane arithmetic, not 32 exec masks held in scalar registers
    auto oob = [&](int e, int k) -> unsigned { return ~(unsigned)((int)(vmask[e] << (31 - k)) >> 31); };
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
        tf2 r1[16], r2[16];                                // residuals of both tiles
        if (!FULL) asm volatile("" : "+v"(vmask[0]), "+v"(vmask[1]));      // (opaque: the per-pixel masks are not to be hoisted out of the round loop)
        cs_u = __builtin_amdgcn_readfirstlane(p.Cout * 4);
        rs_u = __builtin_amdgcn_readfirstlane(p.W * p.Cout * 4);
        asm volatile("" : "+s"(cs_u), "+s"(rs_u));
        // residuals in flight across the barrier and the first transform pass.  Two operands (the skip feature of a decoder stage, 2 of the
        // network's 20 launches): only the upper two tile rows of both now, the lower two behind the first pass, whose registers they take -
        // all 64 values next to the other column tile's accumulators made the allocator spill accumulator tiles INSIDE the main loop
        // (5.6 GB of scratch writes per launch, 2.46 ms instead of 1.5: profiles/r04_run3_pmc_summary_streams1.txt)
        // A block on the image border (!FULL, rare: H or W not a multiple of 16) fetches row by row where the values are used: its 32 masked
        // per-pixel offsets next to 32 residuals in flight pushed the allocator over 256 registers (11 dwords of scratch in every variant)
        if (FULL) {
            if (HAS1) fetch(fullc, rs_r1, r, r1, 0, HAS2 ? 8 : 16);
            if (HAS2) fetch(fullc, rs_r2, r, r2, 0, 8);
        }
        const float bv = p.bias[n0 + 32 * r + c31];
        __syncthreads();
        WTT(3 + 4 * r)
        if (!(FEMASR_WINO_ABL & 16)) {
            // component stride: 16 tile pairs x 32 channels = 4 KiB; three bases keep every read inside the 64-KiB offset field
            int so1 = (16 * 512 + pi * 32 + c31) * 8, so2 = (32 * 512 + pi * 32 + c31) * 8;
            asm volatile("" : "+v"(so1), "+v"(so2));
            const tf2 *src0 = reinterpret_cast<const tf2 *>(Mx) + pi * 32 + c31;
            const tf2 *src1 = reinterpret_cast<const tf2 *>(reinterpret_cast<const char *>(Mx) + so1), *src2 = reinterpret_cast<const tf2 *>(reinterpret_cast<const char *>(Mx) + so2);
            auto mx = [&](int c) -> tf2 { return c < 16 ? src0[c * 512] : (c < 32 ? src1[(c - 16) * 512] : src2[(c - 32) * 512]); };
            tf2 tt[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                at6(mx(0 * 6 + j), mx(1 * 6 + j), mx(2 * 6 + j), mx(3 * 6 + j), mx(4 * 6 + j), mx(5 * 6 + j), tt[0][j], tt[1][j], tt[2][j], tt[3][j]);
                if (FEMASR_WINO_EPI_FENCE && (j & 1)) __builtin_amdgcn_sched_barrier(0);
            }
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
            if (gnp) {      // fp64 from here
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
        WTT(4 + 4 * r)
    };
    // (two per-lane bases, pinned: a wave's pairs 0-7 are the four consecutive components 4 wave .., pair 8 is component
    // 32 + wave / 2; everything else is an immediate offset - left alone the compiler keeps one address register per component)
    // (the opaque values are integer offsets: an opaque POINTER loses its LDS address space and turns the accesses into flat ones)
    int wo0 = (4 * wave) * 4096 + hh * 512 + c31 * 8, wo1 = (32 + (wave >> 1)) * 4096 + hh * 512 + c31 * 8;
    asm volatile("" : "+v"(wo0), "+v"(wo1));
    char *wb0 = reinterpret_cast<char *>(Mx) + wo0, *wb1 = reinterpret_cast<char *>(Mx) + wo1;
    auto write_acc = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (pntl(q) == R) {
                char *dst = q < 8 ? wb0 + (q >> 1) * 4096 : wb1;
#pragma unroll
                for (int e = 0; e < 16; e += 2)      // pair row ((e & 3) + 8 (e >> 2) + 4 hh) / 2, 256 bytes each
                    *reinterpret_cast<tf2 *>(dst + (((e & 3) >> 1) + 4 * (e >> 2)) * 256) = tf2{acc[q][e], acc[q][e + 1]};
            }
        }
    };
    write_acc(std::integral_constant<int, 0>{});          // (ahead of the round loop: these tiles' registers are free from here on)
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        if (r == 1) write_acc(std::integral_constant<int, 1>{});
        WTT(2 + 4 * r)
        if (full) round(std::true_type{}, r); else round(std::false_type{}, r);
        __syncthreads();
        WTT(5 + 4 * r)
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
    WTT_END
}

// 3x3 OIHW -> out[step = ci/8][component 6 i + j][32-column tile][lane][e]: column o = 32 tile + lane%32,
// ci = 8 step + 2 e + lane/32 (the B fragments of the four k-pairs of a step: one 16-byte load per lane, 1 KiB per wave)
__global__ void repack_wino_kernel(const float *__restrict__ in, int O, int I, float *__restrict__ out, size_t total)
{
    const int NT32 = (O + 31) / 32;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        size_t rest = idx >> 8;
        const int ntile = (int)(rest % NT32);
        rest /= NT32;
        const int comp = (int)(rest % 36), step = (int)(rest / 36);
        const int ci = 8 * step + 2 * e + (lane >> 5), o = ntile * 32 + (lane & 31);
        float v = 0.f;
        if (ci < I && o < O) {
            const int i = comp / 6, j = comp - 6 * i;
            const float *gw = in + ((size_t)o * I + ci) * 9;
            float ur[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) ur[x] = wino_g6(i, gw[x], gw[3 + x], gw[6 + x]);
            v = wino_g6(j, ur[0], ur[1], ur[2]);
        }
        out[idx] = v;
    }
}

struct WVariant {
    const char *name;
    void (*kern)(const WinoParams);
    unsigned long long attr_devs;
    size_t attr_lds;
};
#define FEMASR_WINO(PRO, FAST, NRES) { "conv3x3_wino4<2x16x16px x64," #PRO "," #FAST ",res=" #NRES ",waves=8>", conv3x3_wino4_kernel<PRO, FAST, NRES>, 0ull, 0 }
#define FEMASR_WINO3(PRO, FAST) FEMASR_WINO(PRO, FAST, 0), FEMASR_WINO(PRO, FAST, 1), FEMASR_WINO(PRO, FAST, 2)
WVariant g_wv[] = {                               // index = 3 * (prologue form) + residual operands
    FEMASR_WINO3(FEMASR_PRO_NONE, false),
    FEMASR_WINO3(FEMASR_PRO_GN_SILU, false),      // exact SiLU
    FEMASR_WINO3(FEMASR_PRO_GN_SILU, true),       // hardware exp2 / rcp SiLU
};
constexpr int kNumW = sizeof(g_wv) / sizeof(g_wv[0]);

}  // namespace

// Element limits of the Winograd-form kernels (32-bit byte offsets: 2^31 elements per tensor, 2^27 per image).  A handle may LOWER them
// for its planner (femasr_debug_set_wino_limits, include/femasr_hip_debug.h: tests of both sides of the limit at sizes they can allocate).
bool femasr_conv_wino_shape_ok_lim(const femasr_conv_args *a, int log2_total, int log2_image)
{
    const size_t tot = (size_t)1 << log2_total, img = (size_t)1 << log2_image;
    return a->ksz == 3 && a->stride == 1 && a->pad == 1 && !a->up2 && a->act == FEMASR_ACT_NONE && (a->Cin % BK) == 0 && a->Cin <= 1024 &&
           (a->Cout % 64) == 0 && (a->prologue == FEMASR_PRO_NONE || a->prologue == FEMASR_PRO_GN_SILU) &&
           (size_t)a->B * a->H * a->W * a->Cin < tot && (size_t)a->B * a->H * a->W * a->Cout < tot &&
           (size_t)a->H * a->W * a->Cin < img &&          // two images within the 2 GiB range of the input descriptor
           (size_t)a->H * a->W * a->Cout < img &&         // ... and of the output / residual descriptors
           (size_t)36 * a->Cin * a->Cout < ((size_t)1 << 29);
}
bool femasr_conv_wino_shape_ok(const femasr_conv_args *a) { return femasr_conv_wino_shape_ok_lim(a, FEMASR_WINO_LOG2_TOTAL, FEMASR_WINO_LOG2_IMAGE); }
int femasr_conv_wino_variant_count() { return kNumW; }
const char *femasr_conv_wino_variant_name(int v) { return v >= 0 && v < kNumW ? g_wv[v].name : "?"; }
int femasr_conv_wino_gn_tiles(int H, int W) { return ((H + 15) / 16) * ((W + 15) / 16); }

int femasr_conv_wino_launch(hipStream_t s, const femasr_conv_args *a, int *variant_out, double *flops_out)
{
    FEMASR_REQUIRE(a && a->in && a->w_wino && a->bias && a->out && femasr_conv_wino_shape_ok(a), "conv_wino: bad arguments / shape");
    FEMASR_REQUIRE(a->Ho == a->H && a->Wo == a->W, "conv_wino: Ho/Wo mismatch");
    const bool gn = a->prologue == FEMASR_PRO_GN_SILU;
    if (gn) FEMASR_REQUIRE(a->pro_a && a->pro_b, "conv_wino: GN prologue needs a,b");
    FEMASR_REQUIRE(!a->gn_part || femasr_gn_fusable(a->Cout), "conv_wino: gn_part needs 32 | Cout and Cout/32 a power of two <= 32");
    WinoParams p{};
    p.in = a->in; p.u = (const float *)a->w_wino; p.bias = a->bias; p.pro_a = a->pro_a; p.pro_b = a->pro_b;
    p.res1 = a->res1; p.res2 = a->res2; p.out = a->out; p.gn_part = a->gn_part;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
    p.sbX = (a->W + 15) / 16;
    p.sbY = (a->H + 15) / 16;
    p.nsb = a->B * p.sbX * p.sbY;
    p.MB = (p.nsb + 1) / 2;
    p.NB = a->Cout / 64;
    p.nsteps = a->Cin / 8;
    p.NT32 = a->Cout / 32;
    FEMASR_REQUIRE(a->res1 || !a->res2, "conv_wino: res2 without res1");
    FEMASR_REQUIRE(!a->in_add, "conv_wino: in_add is only taken by the x2 form");
    if (flops_out) *flops_out = 2.0 * (double)a->B * a->H * a->W * 9.0 * (double)a->Cin * (double)a->Cout;      // ALGORITHMIC, see below
    const int vi = 3 * (gn ? (a->fast_act ? 2 : 1) : 0) + (a->res1 ? (a->res2 ? 2 : 1) : 0);
    WVariant &v = g_wv[vi];
    const size_t lds = wino_lds_bytes(a->Cin, gn);
    int dev = 0;
    FEMASR_CHECK_HIP(hipGetDevice(&dev));
    {   // the attribute is per device and grows with Cin (GN table): bookkeeping under a lock (several handles / threads share the variants)
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
    // ALGORITHMIC flops (the definition's 9 taps per output pixel, like every other conv launcher); the kernel issues 36/144 of them
    if (flops_out) *flops_out = 2.0 * (double)a->B * a->H * a->W * 9.0 * (double)a->Cin * (double)a->Cout;
    return FEMASR_OK;
}

extern "C" {

#ifdef FEMASR_WINO_TT
int femasr_debug_wino_ttbuf(unsigned long long *dev_buf)      // [blocks][2][64] on the device, zero-filled by the caller
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wi_ttbuf), &dev_buf, sizeof(dev_buf));
}
#endif

size_t femasr_wino_weight_floats(int O, int I) { return (I % 32) == 0 ? (size_t)(I / 8) * 36 * ((O + 31) / 32) * 256 : 0; }

int femasr_repack_oihw_wino(void *stream, const float *in, int O, int I, float *out)
{
    FEMASR_REQUIRE(in && out && O > 0 && I > 0 && (I % 32) == 0, "repack_wino: needs a 3x3 OIHW weight with I %% 32 == 0");
    const size_t total = femasr_wino_weight_floats(O, I);
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(repack_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, O, I, out, total);
    FEMASR_CHECK_HIP(hipGetLastError());
    return FEMASR_OK;
}

}  // extern "C"
