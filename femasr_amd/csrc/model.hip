// model.hip — host side of libfemasr_hip.so: the model handle (weights by state-dict key),
// the workspace planner and the launch schedule of the whole hot path.
//
// Mirrors, op for op, what the reference executes for one call of
//   FeMaSRNet.test / .forward      basicsr/archs/femasr_arch.py:449-479
//   FeMaSRNet.encode_and_decode    femasr_arch.py:311-374
//   MultiScaleEncoder.forward      femasr_arch.py:184-192 (block list built at :150-180)
//   SwinLayers / RSTB / SwinTransformerBlock   femasr_arch.py:114-132; network_swinir.py:442-482,239-279
//   DecoderBlock x3 + out_conv     femasr_arch.py:195-211,267-273,366-369
//   FeMaSRNet.decode_indices       femasr_arch.py:376-385
// but on NHWC activations, with every elementwise / normalisation op folded into the
// neighbouring implicit-GEMM launch (see kernels_conv.hip) — about 300 launches per forward,
// all on the caller's stream, no host synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "common.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

int femasr_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

// every ABI entry runs on the handle's device and restores the caller's current device on return (a model on cuda:1 must
// not change torch's current device for the calling thread)
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int channels_at(int res)   // femasr_arch.py:244-252
{
    switch (res) {
    case 8: case 16: case 32: case 64: return 256;
    case 128: return 128;
    case 256: return 64;
    case 512: return 32;
    default: return -1;
    }
}

int ilog2(int v)
{
    int r = 0;
    while ((1 << (r + 1)) <= v) ++r;
    return r;
}

enum WKind { W_CONV, W_LINEAR, W_VEC, W_TABLE, W_CODEBOOK };

struct WSpec {
    std::string key;
    WKind kind;
    int64_t shape[4];
    int ndim;
    float *dev = nullptr;   // repacked, owned
    void *split = nullptr;  // bf16x3 hi/lo fragments (decoder-side 3x3 convs only), owned
    void *lin3 = nullptr;   // three bf16 planes of a 1x1 / linear weight (femasr_repack_k1_bf16s), owned
    float *up2w = nullptr;  // phase matrices of a nearest-x2 conv (femasr_repack_oihw_up2), owned
    float *wino = nullptr;  // Winograd-domain weights (decoder-side 3x3 convs of single-codebook networks), owned
    bool up2 = false;       // the conv behind nn.Upsample(x2) of an up / decoder block
    bool set = false;
    size_t numel() const { size_t n = 1; for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i]; return n; }
};

// Is the conv with this state-dict key BEHIND every codebook lookup of the network (so that it cannot move a VQ index and may run in a
// cheaper algebraic form)?  One codebook: the whole decoder side, and the LQ encoder's up-blocks (they only make skip features).
// Several codebooks (femasr_arch.py:277-300,330-367): the decoder feeds the later lookups through before_quant_group[q > 0], so only what
// follows the LAST lookup qualifies: after_quant_group[last], decoder_group[i >= stage of the last lookup], out_conv.  A static rule of
// the layer (decode_indices runs decoder_group[0..] of such a network in the direct form too; the oracle applies the same rule).
bool behind_every_lookup(const femasr_config &cfg, int encode_depth, int last_quant_stage, const std::string &k)
{
    const bool dec = k.rfind("decoder_group.", 0) == 0, aq = k.rfind("after_quant_group.", 0) == 0, oc = k.rfind("out_conv", 0) == 0;
    if (cfg.n_codebooks == 1) {
        if (dec || aq || oc) return true;
        const std::string pre = "multiscale_encoder.blocks.";
        return cfg.lq_stage && k.rfind(pre, 0) == 0 && atoi(k.c_str() + pre.size()) > encode_depth;
    }
    if (oc) return true;
    if (aq) return atoi(k.c_str() + 18) == cfg.n_codebooks - 1;
    if (dec) return atoi(k.c_str() + 14) >= last_quant_stage;
    return false;
}

struct T {          // NHWC activation view
    float *p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    double *gn_part = nullptr;   // fused GroupNorm partial moments written by the producing bf16x3 conv (or null)
    int gn_tiles = 0;
    size_t numel() const { return (size_t)B * H * W * C; }
};

// offsets-only first-fit arena over the caller's workspace (plan == execution order, so reuse of a
// freed range by a later launch is safe under stream ordering).
struct Arena {
    char *base = nullptr;
    size_t cap = 0, peak = 0;
    bool dry = true;
    std::map<size_t, size_t> free_;   // offset -> size
    std::map<size_t, size_t> used_;
    void reset(void *b, size_t c, bool d)
    {
        base = d ? (char *)4096 : (char *)b; cap = c; dry = d; peak = 0;   // dry: fake non-null base, offsets only
        free_.clear(); used_.clear();
        free_[0] = d ? ((size_t)1 << 60) : c;
    }
    void *alloc(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second >= bytes) {
                const size_t off = it->first, rest = it->second - bytes;
                free_.erase(it);
                if (rest) free_[off + bytes] = rest;
                used_[off] = bytes;
                peak = std::max(peak, off + bytes);
                return base + off;
            }
        }
        return nullptr;
    }
    void release(void *ptr)
    {
        const size_t off = (size_t)((char *)ptr - base);
        auto it = used_.find(off);
        if (it == used_.end()) return;
        size_t sz = it->second, o = off;
        used_.erase(it);
        auto nx = free_.lower_bound(o);
        if (nx != free_.end() && o + sz == nx->first) { sz += nx->second; nx = free_.erase(nx); }
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == o) { o = pv->first; sz += pv->second; free_.erase(pv); }
        }
        free_[o] = sz;
    }
};

struct ProfRec { int slot; hipEvent_t e0, e1; double flops, bytes; };

}  // namespace

struct femasr_handle {
    femasr_config cfg;
    int scale = 1, max_depth = 0, encode_depth = 0;
    int last_quant_stage = 0;       // decoder stage of the last codebook lookup (behind_every_lookup)
    std::vector<WSpec> specs;
    std::map<std::string, int> index;
    float *cbT[FEMASR_MAX_CODEBOOKS] = {nullptr, nullptr, nullptr}, *ee[FEMASR_MAX_CODEBOOKS] = {nullptr, nullptr, nullptr};
    void *vq_aux[FEMASR_MAX_CODEBOOKS] = {nullptr, nullptr, nullptr};       // bf16 codebook image of the two-pass search
    bool finalized = false;
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    std::vector<double> acc_ms, acc_flops, acc_bytes;
    std::vector<int64_t> acc_n;
    std::vector<std::string> slot_names;
    // sub-batch streams (femasr_set_streams): independent samples run on separate streams so that
    // one sub-batch's kernels fill the tail / HBM-bound phases of the other's
    int decoder_math = 0;   // femasr_set_decoder_math
    int linear_math = 1;    // femasr_set_linear_math: 1 = fp32-grade product on the bf16 matrix pipe, 0 = fp32 MFMA chain
    int wino_log2_total = FEMASR_WINO_LOG2_TOTAL, wino_log2_image = FEMASR_WINO_LOG2_IMAGE;      // femasr_debug_set_wino_limits (planner only)
    int nsub = 1;
    std::vector<hipStream_t> sub_streams;
    std::vector<hipEvent_t> sub_done;
    hipEvent_t ev_fork = nullptr;
    std::map<std::array<int, 4>, std::vector<size_t>> plans;      // workspace plans per (B, H, W, pad_mode)
};

namespace {

enum { SLOT_GN = 0, SLOT_LN, SLOT_ATTN, SLOT_VQ, SLOT_LAYOUT, SLOT_SMALL_COUNT };
const char *kSmallNames[SLOT_SMALL_COUNT] = {"gn_moments", "layernorm", "window_attention", "vq(codebook lookup)",
                                             "pad/crop/gather layout"};

struct Scope {   // event pair around one launch (or a small group of launches)
    femasr_handle *h;
    ProfRec rec{};
    bool on;
    hipStream_t st;
    Scope(femasr_handle *hh, hipStream_t stream, bool dry, int slot, double flops, double bytes)
        : h(hh), on(hh->prof && !dry), st(stream)
    {
        if (!on) return;
        auto get = [&]() {
            if (h->pool_used == h->pool.size()) {
                hipEvent_t e;
                (void)hipEventCreate(&e);
                h->pool.push_back(e);
            }
            return h->pool[h->pool_used++];
        };
        rec.slot = slot; rec.flops = flops; rec.bytes = bytes;
        rec.e0 = get(); rec.e1 = get();
        (void)hipEventRecord(rec.e0, st);
    }
    void set_slot(int s) { rec.slot = s; }
    void set_flops(double f) { rec.flops = f; }
    void set_bytes(double b) { rec.bytes = b; }
    ~Scope()
    {
        if (!on) return;
        (void)hipEventRecord(rec.e1, st);
        h->recs.push_back(rec);
    }
};

void add_conv(femasr_handle *h, const std::string &p, int cin, int cout, int k, bool up2 = false)
{
    WSpec w; w.key = p + ".weight"; w.kind = W_CONV; w.ndim = 4;
    w.shape[0] = cout; w.shape[1] = cin; w.shape[2] = k; w.shape[3] = k;
    w.up2 = up2;
    h->specs.push_back(w);
    WSpec b; b.key = p + ".bias"; b.kind = W_VEC; b.ndim = 1; b.shape[0] = cout;
    h->specs.push_back(b);
}
void add_linear(femasr_handle *h, const std::string &p, int cin, int cout)
{
    WSpec w; w.key = p + ".weight"; w.kind = W_LINEAR; w.ndim = 2; w.shape[0] = cout; w.shape[1] = cin;
    h->specs.push_back(w);
    WSpec b; b.key = p + ".bias"; b.kind = W_VEC; b.ndim = 1; b.shape[0] = cout;
    h->specs.push_back(b);
}
void add_norm(femasr_handle *h, const std::string &p, int c)
{
    for (const char *leaf : {".weight", ".bias"}) {
        WSpec w; w.key = p + leaf; w.kind = W_VEC; w.ndim = 1; w.shape[0] = c;
        h->specs.push_back(w);
    }
}
void add_resblock(femasr_handle *h, const std::string &p, int c)
{
    add_norm(h, p + ".conv.0.norm", c);
    add_conv(h, p + ".conv.2", c, c, 3);
    add_norm(h, p + ".conv.3.norm", c);
    add_conv(h, p + ".conv.5", c, c, 3);
}

int build_specs(femasr_handle *h)
{
    const femasr_config &c = h->cfg;
    const std::string enc = "multiscale_encoder";
    int res = c.gt_resolution / h->scale;
    FEMASR_REQUIRE(channels_at(res) > 0, "unsupported encoder input resolution %d", res);
    add_conv(h, enc + ".in_conv", c.in_channel, channels_at(res), 4);
    int bi = 0;
    for (int i = 0; i < h->encode_depth; ++i, ++bi) {
        const int ic = channels_at(res), oc = channels_at(res / 2);
        FEMASR_REQUIRE(ic > 0 && oc > 0, "unsupported resolution %d", res);
        const std::string p = enc + ".blocks." + std::to_string(bi);
        add_conv(h, p + ".0", ic, oc, 3);
        add_resblock(h, p + ".1", oc);
        add_resblock(h, p + ".2", oc);
        res /= 2;
    }
    if (c.lq_stage) {
        FEMASR_REQUIRE(channels_at(res) == 256, "Swin stage needs 256 channels at res %d", res);
        const std::string sp = enc + ".blocks." + std::to_string(bi++);
        for (int r = 0; r < 4; ++r) {
            const std::string rp = sp + ".swin_blks." + std::to_string(r);
            for (int k = 0; k < 6; ++k) {
                const std::string bp = rp + ".residual_group.blocks." + std::to_string(k);
                add_norm(h, bp + ".norm1", 256);
                WSpec tb; tb.key = bp + ".attn.relative_position_bias_table"; tb.kind = W_TABLE; tb.ndim = 2;
                tb.shape[0] = 225; tb.shape[1] = 8;
                h->specs.push_back(tb);
                add_linear(h, bp + ".attn.qkv", 256, 768);
                add_linear(h, bp + ".attn.proj", 256, 256);
                add_norm(h, bp + ".norm2", 256);
                add_linear(h, bp + ".mlp.fc1", 256, 1024);
                add_linear(h, bp + ".mlp.fc2", 1024, 256);
            }
            add_conv(h, rp + ".conv", 256, 256, 3);
        }
        for (int u = 0; u < 2; ++u, ++bi) {
            const int ic = channels_at(res), oc = channels_at(res * 2);
            const std::string p = enc + ".blocks." + std::to_string(bi);
            add_conv(h, p + ".1", ic, oc, 3, true);
            add_resblock(h, p + ".2", oc);
            add_resblock(h, p + ".3", oc);
            res *= 2;
        }
    }
    int out_ch = 0;
    for (int i = 0; i < h->max_depth; ++i) {
        const int r = c.gt_resolution / (1 << h->max_depth) * (1 << i);
        const int ic = channels_at(r), oc = channels_at(r * 2);
        FEMASR_REQUIRE(ic > 0 && oc > 0, "unsupported decoder resolution %d", r);
        const std::string p = "decoder_group." + std::to_string(i) + ".block";
        add_conv(h, p + ".1", ic, oc, 3, true);
        add_resblock(h, p + ".2", oc);
        add_resblock(h, p + ".3", oc);
        out_ch = oc;
    }
    add_conv(h, "out_conv", out_ch, 3, 3);
    // multi-scale vector quantisers (femasr_arch.py:277-300): group q > 0 quantises cat(enc_feat, prev_dec_feat) and its
    // CombineQuantBlock convolves cat(z_quant, resized previous z_quant)
    for (int q = 0; q < c.n_codebooks; ++q) {
        const std::string qs = std::to_string(q);
        WSpec cb; cb.key = "quantize_group." + qs + ".embedding.weight"; cb.kind = W_CODEBOOK; cb.ndim = 2;
        cb.shape[0] = c.n_e[q]; cb.shape[1] = c.e_dim[q];
        h->specs.push_back(cb);
        const int qc = channels_at(c.codebook_scale[q]);
        FEMASR_REQUIRE(qc > 0, "unsupported codebook scale %d", c.codebook_scale[q]);
        add_conv(h, "before_quant_group." + qs, q == 0 ? qc : 2 * qc, c.e_dim[q], 1);
        add_conv(h, "after_quant_group." + qs + ".conv", q == 0 ? c.e_dim[0] : c.e_dim[q - 1] + c.e_dim[q], qc, 3);
    }
    for (size_t i = 0; i < h->specs.size(); ++i) h->index[h->specs[i].key] = (int)i;
    return FEMASR_OK;
}

// ---------------------------------------------------------------- forward-time helpers
struct Ctx {
    femasr_handle *h;
    Arena *arena;
    hipStream_t stream;
    int rc = 0;
    // femasr_forward_u8: the first / last kernel of the forward read / write uint8 HWC images directly (bit 0: input, bit 1: output)
    int io_u8 = 0, swap_rb = 0;
    bool dry() const { return arena->dry; }
    hipStream_t s() const { return stream; }

    const float *Wt(const std::string &key)
    {
        auto it = h->index.find(key);
        if (it == h->index.end()) { if (!rc) rc = femasr_set_error(FEMASR_ERR_WEIGHT, "internal: unknown weight %s", key.c_str()); return nullptr; }
        return h->specs[it->second].dev;
    }
    float *alloc_f(size_t n)
    {
        void *p = arena->alloc(n * sizeof(float));
        if (!p && !rc) rc = femasr_set_error(FEMASR_ERR_WORKSPACE, "workspace too small (need > %zu bytes)", arena->cap);
        return (float *)p;
    }
    T alloc_t(int B, int H, int W, int C)
    {
        T t; t.B = B; t.H = H; t.W = W; t.C = C;
        t.p = alloc_f(t.numel());
        return t;
    }
    void release(void *p) { if (p) arena->release(p); }
    void release(T &t) { release(t.p); t.p = nullptr; }

    struct ConvOpt {
        int ksz = 3, stride = 1, pad = 1, up2 = 0, act = 0;
        int pro = FEMASR_PRO_NONE;
        const float *pa = nullptr, *pb = nullptr, *pc = nullptr;
        const float *res1 = nullptr, *res2 = nullptr;
        const float *in_add = nullptr;   // second input, added while staging: only when up2_wino_ok() said the x2 Winograd-type form will run
        bool lowp = false;       // behind the VQ lookup: may use the bf16x3 path when the handle opts in
        bool want_gn = false;    // the output feeds a GroupNorm: let a bf16x3 conv emit its partial moments
    };
    T conv(const T &x, const std::string &prefix, int cout, const ConvOpt &o)
    {
        const int Hv = o.up2 ? 2 * x.H : x.H, Wv = o.up2 ? 2 * x.W : x.W;
        const int Ho = (Hv + 2 * o.pad - o.ksz) / o.stride + 1, Wo = (Wv + 2 * o.pad - o.ksz) / o.stride + 1;
        T y = alloc_t(x.B, Ho, Wo, cout);
        // The args struct is populated (shapes; pointers may be null in the dry run) BEFORE the planning decisions, and the
        // same eligibility helpers decide in the dry and in the real run, so both plan identical buffers.
        femasr_conv_args a{};
        a.in = x.p; a.B = x.B; a.H = x.H; a.W = x.W; a.Cin = x.C;
        a.Cout = cout; a.ksz = o.ksz; a.stride = o.stride; a.pad = o.pad; a.up2 = o.up2;
        a.prologue = o.pro; a.pro_a = o.pa; a.pro_b = o.pb; a.pro_c = o.pc;
        a.act = o.act; a.res1 = o.res1; a.res2 = o.res2; a.out = y.p; a.Ho = Ho; a.Wo = Wo;
        a.in_add = o.in_add;
        const void *split = nullptr;
        // (several codebooks: the decoder feeds the later lookups through before_quant_group[q > 0]: only the convs that follow the LAST
        // lookup may take the bf16x3 / Winograd forms - behind_every_lookup)
        const bool behind = o.lowp && behind_every_lookup(h->cfg, h->encode_depth, h->last_quant_stage, prefix);
        if (behind && h->decoder_math == 1) {
            auto it = h->index.find(prefix + ".weight");
            if (it != h->index.end()) split = h->specs[it->second].split;
        }
        const bool lowp_on = split != nullptr && cout > 4 && femasr_conv_bf16x3_shape_ok(&a);      // out_conv: exact VALU kernel in both modes
        // exact-fp32 mode: convs behind every codebook lookup run in the Winograd F(4x4,3x3) form
        // (they cannot move a VQ index; oracle: OracleNet.wino).  decoder_math 2 = 'fp32_direct' keeps the direct form; 0 runs the
        // SiLU of their GroupNorm prologue on the hardware exp2 / rcp units, 3 = 'fp32_strict' keeps it IEEE-exact (== oracle).
        const bool wino_on = !lowp_on && behind && (h->decoder_math == 0 || h->decoder_math == 3) &&
                             (o.up2 ? femasr_conv_wino_up2_shape_ok_lim(&a, h->wino_log2_total, h->wino_log2_image)
                                    : femasr_conv_wino_shape_ok_lim(&a, h->wino_log2_total, h->wino_log2_image));
        // linear_math 1 (round 6): a 3x3 stride-1 conv in FRONT of a lookup (encoder ResBlocks, the conv behind every RSTB) runs as the
        // split-bf16 GEMM over K = 9 Cin - the arithmetic of the linear layers (oracle: conv3x3_bf16s).  Its GroupNorm + SiLU prologue
        // becomes a pass of its own (the GEMM takes plain rows), and it emits no GroupNorm partials: the next GroupNorm runs the
        // stand-alone moments kernel, which gives the same coefficients bit for bit.
        bool split3 = false;
        {
            femasr_conv_args q = a;
            q.prologue = FEMASR_PRO_NONE;
            split3 = h->linear_math == 1 && !behind && o.ksz == 3 && !lowp_on && !wino_on && !o.in_add &&
                     (o.pro == FEMASR_PRO_NONE || o.pro == FEMASR_PRO_GN_SILU) && femasr_conv3x3_bf16s_shape_ok(&q);
        }
        float *act_tmp = nullptr;
        if (split3 && o.pro == FEMASR_PRO_GN_SILU) act_tmp = alloc_f(x.numel());
        const bool gn_ok = split3 ? false : lowp_on ? (cout % 32 == 0 && cout / 32 <= 8 && ((cout / 32) & (cout / 32 - 1)) == 0)
                                   : (femasr_conv_halo_eligible(&a) && femasr_gn_fusable(cout));
        if (o.want_gn && gn_ok) {
            // exact x2 convs (phase filters) emit one partial per half-resolution tile and phase
            // (a Winograd conv emits one partial per 16x16-pixel sub-block, orc_gn_coeffs mode 2)
            y.gn_tiles = wino_on ? femasr_conv_wino_gn_tiles(Ho, Wo)
                         : ((o.up2 && !lowp_on) ? 4 * ((x.H + 7) / 8) * ((x.W + 15) / 16) : ((Ho + 7) / 8) * ((Wo + 15) / 16));
            y.gn_part = (double *)arena->alloc((size_t)x.B * y.gn_tiles * 32 * 2 * sizeof(double));
            if (!y.gn_part && !rc) rc = femasr_set_error(FEMASR_ERR_WORKSPACE, "workspace too small");
        }
        if (rc || dry()) { release(act_tmp); return y; }
        a.w = Wt(prefix + ".weight"); a.bias = Wt(prefix + ".bias");
        a.gn_part = y.gn_part;
        if (o.up2) {
            auto it = h->index.find(prefix + ".weight");
            if (it != h->index.end()) a.w_up2 = h->specs[it->second].up2w;
        }
        if (rc) { release(act_tmp); return y; }
        if (split3 && act_tmp) {       // GroupNorm-apply + SiLU as its own pass (exact SiLU: these convs feed the lookup)
            Scope sa(h, s(), dry(), SLOT_GN, 0.0, (double)x.numel() * 8.0);
            const int ra = femasr_gn_silu_apply(s(), x.p, x.B, x.H, x.W, x.C, o.pa, o.pb, act_tmp);
            if (ra && !rc) rc = ra;
            a.in = act_tmp;
            a.prologue = FEMASR_PRO_NONE; a.pro_a = nullptr; a.pro_b = nullptr;
        }
        Scope sc(h, s(), dry(), 0, 0.0, 0.0);
        int variant = 0; double flops = 0;
        int r;
        const float *wino_w = nullptr;
        if (wino_on) {
            auto it = h->index.find(prefix + ".weight");
            if (it != h->index.end()) wino_w = h->specs[it->second].wino;
        }
        if (o.in_add && !(wino_on && o.up2)) {
            rc = femasr_set_error(FEMASR_ERR_INVALID, "conv %s: a second input was scheduled for a conv that does not run in the x2 Winograd-type form", prefix.c_str());
            return y;
        }
        if (wino_on && !wino_w) {
            // the plan sized gn_part for the Winograd form (one partial per 16x16 sub-block); the direct kernels write 2-4x as many:
            // never fall through to them - set_weight packs the Winograd weights for exactly these shapes
            rc = femasr_set_error(FEMASR_ERR_WEIGHT, "conv %s: planned in the Winograd form but its weights were not packed for it", prefix.c_str());
            return y;
        }
        const void *lin3 = nullptr;
        if (h->linear_math == 1 && (split3 || (o.ksz == 1 && femasr_gemm_bf16s_shape_ok(&a)))) {
            auto it = h->index.find(prefix + ".weight");
            if (it != h->index.end()) lin3 = h->specs[it->second].lin3;
        }
        if (split3 && !lin3) {
            rc = femasr_set_error(FEMASR_ERR_WEIGHT, "conv %s: planned as the split-bf16 GEMM but its weight planes were not packed", prefix.c_str());
            release(act_tmp);
            return y;
        }

        if (lin3) {
            r = femasr_gemm_bf16s_launch(s(), &a, lin3, &variant, &flops);
            variant += femasr_conv_variant_count() + femasr_conv_bf16x3_variant_count() + femasr_conv_wino_variant_count() + 1;
        } else if (lowp_on) {
            a.w_bf16x3 = split;
            r = femasr_conv_bf16x3_launch(s(), &a, &variant, &flops);
            variant += femasr_conv_variant_count();
        } else if (wino_w) {
            a.w_wino = wino_w;
            a.fast_act = h->decoder_math == 0 ? 1 : 0;
            if (o.up2) {
                r = femasr_conv_wino_up2_launch(s(), &a, &flops);
                variant = femasr_conv_wino_variant_count();          // the slot behind the F(4x4,3x3) variants
            } else {
                r = femasr_conv_wino_launch(s(), &a, &variant, &flops);
            }
            variant += femasr_conv_variant_count() + femasr_conv_bf16x3_variant_count();
        } else {
            r = femasr_conv2d_launch(s(), &a, nullptr, &variant, &flops);
        }
        sc.set_slot(SLOT_SMALL_COUNT + variant);
        sc.set_flops(flops);
        // algorithmic bytes of the launch: every operand once - input (+ a second input), residuals, output, weights
        sc.set_bytes(4.0 * ((double)x.numel() * (o.in_add ? 2 : 1) + (double)y.numel() * (1 + (o.res1 ? 1 : 0) + (o.res2 ? 1 : 0)) +
                            (double)cout * x.C * o.ksz * o.ksz + cout));
        if (r && !rc) rc = r;
        release(act_tmp);
        return y;
    }

    // GroupNorm moments -> (a,b); returns a pointer pair inside one allocation
    float *gn(T &x, const std::string &norm_prefix)
    {
        float *ab = alloc_f((size_t)2 * x.B * x.C);
        if (x.gn_part) {         // moments were accumulated by the producing halo conv's epilogue: finalize only
            if (!rc && !dry()) {
                Scope sc(h, s(), dry(), SLOT_GN, 0.0, (double)x.B * x.gn_tiles * 32 * 16.0);
                const int r = femasr_gn_coeffs_from_partials(s(), x.gn_part, x.B, x.gn_tiles, x.H, x.W, x.C, 32, Wt(norm_prefix + ".weight"),
                                                             Wt(norm_prefix + ".bias"), 1e-6f, ab, ab + (size_t)x.B * x.C);
                if (r && !rc) rc = r;
            }
            release(x.gn_part);
            x.gn_part = nullptr;
            return ab;
        }
        void *scratch = arena->alloc(femasr_gn_scratch_bytes(x.B, x.H, x.W, x.C, 32));
        if (!scratch && !rc) rc = femasr_set_error(FEMASR_ERR_WORKSPACE, "workspace too small");
        if (!rc && !dry()) {
            Scope sc(h, s(), dry(), SLOT_GN, 0.0, (double)x.numel() * 4.0);
            const int r = femasr_gn_coeffs(s(), x.p, x.B, x.H, x.W, x.C, 32, Wt(norm_prefix + ".weight"), Wt(norm_prefix + ".bias"),
                                           1e-6f, ab, ab + (size_t)x.B * x.C, scratch);
            if (r && !rc) rc = r;
        }
        release(scratch);
        return ab;
    }

    // fema_utils.py:65-84 (+ optional fused `x + enc_feats[i]`, femasr_arch.py:361-362)
    T resblock(T x, const std::string &p, const float *res2, bool free_x, bool lowp = false, bool out_gn = false)
    {
        const size_t bc = (size_t)x.B * x.C;
        float *ab = gn(x, p + ".conv.0.norm");
        ConvOpt o1; o1.pro = FEMASR_PRO_GN_SILU; o1.pa = ab; o1.pb = ab ? ab + bc : nullptr; o1.lowp = lowp; o1.want_gn = true;
        T u = conv(x, p + ".conv.2", x.C, o1);
        release(ab);
        float *ab2 = gn(u, p + ".conv.3.norm");
        ConvOpt o2; o2.pro = FEMASR_PRO_GN_SILU; o2.pa = ab2; o2.pb = ab2 ? ab2 + bc : nullptr;
        o2.res1 = x.p; o2.res2 = res2; o2.lowp = lowp; o2.want_gn = out_gn;
        T y = conv(u, p + ".conv.5", x.C, o2);
        release(ab2);
        release(u);
        if (free_x) release(x);
        return y;
    }

    // network_swinir.py:239-279 on tokens (1, rows, 1, C)
    T swin_block(const T &y, int B, int H, int W, const std::string &bp, int shift)
    {
        const int rows = B * H * W, C = 256;
        auto ln = [&](const T &t, const std::string &np) {       // normalised tokens, materialised once (read by DMA in the GEMM)
            T o = alloc_t(1, rows, 1, C);
            if (rc || dry()) return o;
            Scope sc(h, s(), dry(), SLOT_LN, 0.0, (double)t.numel() * 8.0);
            const int r = femasr_layernorm(s(), t.p, rows, C, Wt(np + ".weight"), Wt(np + ".bias"), 1e-5f, o.p);
            if (r && !rc) rc = r;
            return o;
        };
        T n1 = ln(y, bp + ".norm1");
        ConvOpt oq; oq.ksz = 1; oq.pad = 0;
        T qkv = conv(n1, bp + ".attn.qkv", 3 * C, oq);
        release(n1);
        T att = alloc_t(1, rows, 1, C);
        if (!rc && !dry()) {
            Scope sc(h, s(), dry(), SLOT_ATTN, 4.0 * (double)rows * 64.0 * C, (double)rows * C * 16.0);
            const int r = femasr_window_attention(s(), qkv.p, B, H, W, C, 8, shift, Wt(bp + ".attn.relative_position_bias_table"), att.p);
            if (r && !rc) rc = r;
        }
        release(qkv);
        ConvOpt op; op.ksz = 1; op.pad = 0; op.res1 = y.p;
        T y1 = conv(att, bp + ".attn.proj", C, op);
        release(att);
        T n2 = ln(y1, bp + ".norm2");
        // (one kernel for fc1 + GELU + fc2 + residual was built in round 4 - bit-identical, 18 % slower than the two launches,
        // profiles/r04_mlp_fused.txt - and removed in round 6; producer-side bf16 planes for qkv / fc1 / fc2 likewise, kernels_gemm_bf16.hip)
        ConvOpt o1; o1.ksz = 1; o1.pad = 0; o1.act = FEMASR_ACT_GELU;
        T hdn = conv(n2, bp + ".mlp.fc1", 4 * C, o1);
        release(n2);
        ConvOpt o2; o2.ksz = 1; o2.pad = 0; o2.res1 = y1.p;
        T y2 = conv(hdn, bp + ".mlp.fc2", C, o2);
        release(hdn);
        release(y1);
        return y2;
    }

    // femasr_arch.py:114-132 + network_swinir.py:442-482; consumes x
    T swin_layers(T x, const std::string &sp)
    {
        const int B = x.B, H = x.H, W = x.W, C = x.C;
        for (int r = 0; r < 4; ++r) {
            const std::string rp = sp + ".swin_blks." + std::to_string(r);
            T y; y.p = x.p; y.B = 1; y.H = B * H * W; y.W = 1; y.C = C;
            for (int k = 0; k < 6; ++k) {
                T yn = swin_block(y, B, H, W, rp + ".residual_group.blocks." + std::to_string(k), (k & 1) ? 4 : 0);
                if (y.p != x.p) release(y);
                y = yn;
            }
            T yi; yi.p = y.p; yi.B = B; yi.H = H; yi.W = W; yi.C = C;
            ConvOpt o; o.res1 = x.p;
            T nx = conv(yi, rp + ".conv", C, o);
            release(y);
            release(x);
            x = nx;
        }
        return x;
    }

    // Will the x2 conv of a decoder stage with this input run in the Winograd-type form (the same test conv() makes)?  Then the skip
    // feature of that stage is added by ITS staging (in_add) instead of by the previous stage's last epilogue (a second residual operand).
    bool up2_wino_ok(const std::string &prefix, int B, int H, int W, int Cin, int Cout) const
    {
        if (!behind_every_lookup(h->cfg, h->encode_depth, h->last_quant_stage, prefix)) return false;
        femasr_conv_args a{};
        a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ksz = 3; a.stride = 1; a.pad = 1; a.up2 = 1;
        a.prologue = FEMASR_PRO_NONE; a.act = FEMASR_ACT_NONE; a.Ho = 2 * H; a.Wo = 2 * W;
        return (h->decoder_math == 0 || h->decoder_math == 3) && femasr_conv_wino_up2_shape_ok_lim(&a, h->wino_log2_total, h->wino_log2_image);
    }

    T up_block(const T &x, const std::string &p, int cout, const float *res2_last, bool lowp = false, const float *in_add = nullptr)   // Upsample x2 -> conv -> RB -> RB
    {
        ConvOpt o; o.up2 = 1; o.lowp = lowp; o.want_gn = true; o.in_add = in_add;
        T c = conv(x, p + ".1", cout, o);
        T r1 = resblock(c, p + ".2", nullptr, true, lowp, true);
        return resblock(r1, p + ".3", res2_last, true, lowp, false);
    }
};

// Shapes of one call (reference geometry; femasr_arch.py:449-468 test(), :470-479 forward(), encoder strides :150-180)
struct Geometry {
    int Hp, Wp;            // input after test()'s mirror pad
    int eh, ew;            // encoder bottom (codebook scale 0) feature map
    int out_h, out_w;      // decoder output
    int crop_h, crop_w;    // what the caller receives
    int nq;                // number of index maps
    int qh[FEMASR_MAX_CODEBOOKS], qw[FEMASR_MAX_CODEBOOKS];
};

bool quant_at(const femasr_handle *h, int i, int *which)
{
    const int cur_res = h->cfg.gt_resolution / (1 << h->max_depth) * (1 << i);
    for (int q = 0; q < h->cfg.n_codebooks; ++q)
        if (h->cfg.codebook_scale[q] == cur_res) { if (which) *which = q; return true; }
    return false;
}

int plan_geometry(const femasr_handle *h, int H, int W, int pad_mode, Geometry *g)
{
    g->Hp = H; g->Wp = W;
    if (pad_mode) {
        const int wsz = 8 / h->scale * 8;
        g->Hp = (H / wsz + 1) * wsz;
        g->Wp = (W / wsz + 1) * wsz;
        if (h->cfg.lq_stage) {
            FEMASR_REQUIRE(g->Hp - H <= H && g->Wp - W <= W, "test(): image %dx%d smaller than its mirror pad", H, W);
        } else {            // torch.cat([x, flip(x)])[..., :h+pad] holds at most 2h rows: the reference silently truncates the pad
            g->Hp = std::min(g->Hp, 2 * H);
            g->Wp = std::min(g->Wp, 2 * W);
        }
    }
    FEMASR_REQUIRE(g->Hp >= 2 && g->Wp >= 2, "input too small");
    // in_conv (k4, p1) gives Hp-1; each stride-2 conv gives floor((h-1)/2)+1
    int eh = g->Hp - 1, ew = g->Wp - 1;
    for (int i = 0; i < h->encode_depth; ++i) { eh = (eh - 1) / 2 + 1; ew = (ew - 1) / 2 + 1; }
    FEMASR_REQUIRE(eh >= 1 && ew >= 1, "input too small");
    if (h->cfg.lq_stage) FEMASR_REQUIRE(eh % 8 == 0 && ew % 8 == 0, "Swin stage needs a feature map divisible by 8, got %dx%d", eh, ew);
    g->eh = eh; g->ew = ew;
    g->out_h = eh << h->max_depth; g->out_w = ew << h->max_depth;
    const int s_out = pad_mode ? h->scale : 1;
    g->crop_h = pad_mode ? std::min(H * s_out, g->out_h) : g->out_h;      // `output[..., :h*sf, :w*sf]` (a slice never grows)
    g->crop_w = pad_mode ? std::min(W * s_out, g->out_w) : g->out_w;
    g->nq = 0;
    for (int i = 0; i < h->max_depth; ++i) {
        int q;
        if (quant_at(h, i, &q)) { g->qh[g->nq] = eh << i; g->qw[g->nq] = ew << i; ++g->nq; }
    }
    return FEMASR_OK;
}

// decoder_group[first..] + out_conv + crop (femasr_arch.py:366-369); quantisation steps of the later scales happen inside
int run_tail(Ctx &c, T x, std::vector<T> &feats, bool fuse_skip, bool with_encoder, int64_t *const *idx_out, float *out_nchw,
             int crop_h, int crop_w)
{
    femasr_handle *h = c.h;
    const femasr_config &cfg = h->cfg;
    T prev_dec, prev_q, pending_add;      // pending_add: the skip feature the NEXT stage's x2 conv adds to its input
    int nq_done = 0;
    for (int i = 0; i < h->max_depth; ++i) {
        const int r = cfg.gt_resolution / (1 << h->max_depth) * (1 << i);
        int q = 0;
        if (with_encoder && quant_at(h, i, &q)) {       // femasr_arch.py:332-359
            const std::string qs = std::to_string(q);
            T zin = feats[i];
            if (prev_dec.p || (c.dry() && i > 0)) {     // cat((enc_feats[i], prev_dec_feat), dim=1)
                if (feats[i].H != x.H || feats[i].W != x.W) {
                    if (!c.rc) c.rc = femasr_set_error(FEMASR_ERR_INVALID, "multi-codebook: encoder feature %dx%d and decoder feature %dx%d differ", feats[i].H, feats[i].W, x.H, x.W);
                    return c.rc;
                }
                zin = c.alloc_t(x.B, x.H, x.W, feats[i].C + x.C);
                if (!c.rc && !c.dry()) {
                    Scope sc(h, c.s(), c.dry(), SLOT_LAYOUT, 0.0, (double)zin.numel() * 8.0);
                    const int rr = femasr_concat_resize(c.s(), feats[i].p, feats[i].C, x.p, x.H, x.W, x.C, x.B, x.H, x.W, zin.p);
                    if (rr && !c.rc) c.rc = rr;
                }
                c.release(feats[i]);
                c.release(x);
            }
            Ctx::ConvOpt oq; oq.ksz = 1; oq.pad = 0;
            T z = c.conv(zin, "before_quant_group." + qs, cfg.e_dim[q], oq);
            c.release(zin);
            const int64_t M = (int64_t)z.B * z.H * z.W;
            T zq = c.alloc_t(z.B, z.H, z.W, cfg.e_dim[q]);
            int64_t *idx = idx_out ? idx_out[nq_done] : nullptr, *idx_tmp = nullptr;
            if (!idx) idx = idx_tmp = (int64_t *)c.arena->alloc((size_t)M * 8);
            {
                float *scratch = c.alloc_f(femasr_vq_scratch_bytes(M, cfg.n_e[q]) / sizeof(float));
                if (!c.rc && !c.dry()) {
                    Scope sc(h, c.s(), c.dry(), SLOT_VQ, 2.0 * (double)M * cfg.n_e[q] * cfg.e_dim[q],
                             (double)M * cfg.e_dim[q] * 8.0 + (double)cfg.n_e[q] * cfg.e_dim[q] * 4.0 + M * 8.0);
                    const float *cbw = c.Wt("quantize_group." + qs + ".embedding.weight");
                    const int rr = h->vq_aux[q]
                        ? femasr_vq_twopass(c.s(), z.p, M, cfg.e_dim[q], cbw, h->vq_aux[q], h->ee[q], cfg.n_e[q], idx, zq.p, scratch)
                        : femasr_vq(c.s(), z.p, M, cfg.e_dim[q], cbw, h->cbT[q], h->ee[q], cfg.n_e[q], idx, zq.p, scratch);
                    if (rr && !c.rc) c.rc = rr;
                }
                c.release(scratch);
            }
            if (idx_tmp) c.release(idx_tmp);
            ++nq_done;
            T qv = cfg.use_quantize ? zq : z;
            c.release(cfg.use_quantize ? z : zq);
            T ain = qv;
            if (prev_q.p || (c.dry() && nq_done > 1)) {     // CombineQuantBlock: cat((z_quant, interpolate(prev_quant, nearest)), 1)
                ain = c.alloc_t(qv.B, qv.H, qv.W, qv.C + prev_q.C);
                if (!c.rc && !c.dry()) {
                    Scope sc(h, c.s(), c.dry(), SLOT_LAYOUT, 0.0, (double)ain.numel() * 8.0);
                    const int rr = femasr_concat_resize(c.s(), qv.p, qv.C, prev_q.p, prev_q.H, prev_q.W, prev_q.C, qv.B, qv.H, qv.W, ain.p);
                    if (rr && !c.rc) c.rc = rr;
                }
                c.release(prev_q);
            }
            Ctx::ConvOpt oa; oa.lowp = true;
            x = c.conv(ain, "after_quant_group." + qs + ".conv", channels_at(cfg.codebook_scale[q]), oa);
            if (ain.p != qv.p) c.release(ain);
            bool later = false;                             // is this z_quant combined into a later scale?
            for (int k = i + 1; k < h->max_depth; ++k) later = later || quant_at(h, k, nullptr);
            if (later) prev_q = qv; else c.release(qv);
        }
        // `x = x + enc_feats[i+1]` of the NEXT iteration (femasr_arch.py:361-362): added by the next stage's x2 conv while it stages its
        // input when that conv runs in the Winograd-type form (same fp32 add, no third operand in this block's last epilogue - the
        // two-residual epilogue was the slowest instantiation of the F(4x4) kernel); otherwise folded into this block's last epilogue
        const bool next_skip = with_encoder && fuse_skip && i + 1 < h->max_depth && !quant_at(h, i + 1, nullptr);
        const int cout_i = channels_at(r * 2);
        const bool skip_by_next = next_skip && c.up2_wino_ok("decoder_group." + std::to_string(i + 1) + ".block.1", x.B, 2 * x.H, 2 * x.W, cout_i, channels_at(r * 4));
        const float *skip = (next_skip && !skip_by_next) ? feats[i + 1].p : nullptr;
        T y = c.up_block(x, "decoder_group." + std::to_string(i) + ".block", cout_i, skip, true, pending_add.p);
        c.release(x);
        if (pending_add.p) { c.release(pending_add); pending_add = T{}; }
        if (next_skip) { if (skip_by_next) pending_add = feats[i + 1]; else c.release(feats[i + 1]); }
        x = y;
        prev_dec = x;
    }
    Ctx::ConvOpt oo; oo.lowp = true;
    T img = c.conv(x, "out_conv", 3, oo);
    c.release(x);
    if (!c.rc && !c.dry()) {
        Scope sc(h, c.s(), c.dry(), SLOT_LAYOUT, 0.0, (double)img.B * 3.0 * crop_h * crop_w * 8.0);
        const int r = (c.io_u8 & 2) ? femasr_crop_nhwc_to_u8hwc(c.s(), img.p, img.B, img.H, img.W, crop_h, crop_w, c.swap_rb, (uint8_t *)out_nchw)
                                    : femasr_crop_nhwc_to_nchw(c.s(), img.p, img.B, img.H, img.W, 3, crop_h, crop_w, out_nchw);
        if (r && !c.rc) c.rc = r;
    }
    c.release(img);
    return c.rc;
}

int run_forward(femasr_handle *h, Arena *arena, hipStream_t stream, const float *in_nchw, int B, int H, int W, int pad_mode,
                float *out_nchw, int64_t *const *idx_out, int io_u8 = 0, int swap_rb = 0)
{
    const femasr_config &cfg = h->cfg;
    Ctx c{h, arena, stream};
    c.io_u8 = io_u8; c.swap_rb = swap_rb;
    Geometry g;
    int rc = plan_geometry(h, H, W, pad_mode, &g);
    if (rc) return rc;
    const bool fuse_skip = cfg.lq_stage && cfg.use_residual;
    auto needed = [&](int i) {        // is enc_feats[i] read by the decoder side?
        return i == 0 || quant_at(h, i, nullptr) || fuse_skip;
    };

    T x0 = c.alloc_t(B, g.Hp, g.Wp, cfg.in_channel);
    if (!c.rc && !c.dry()) {
        Scope sc(h, c.s(), c.dry(), SLOT_LAYOUT, 0.0, (double)x0.numel() * 8.0);
        const int r = (c.io_u8 & 1) ? femasr_pad_u8hwc_to_nhwc(c.s(), (const uint8_t *)in_nchw, B, H, W, c.swap_rb, g.Hp, g.Wp, x0.p)
                                    : femasr_pad_nchw_to_nhwc(c.s(), in_nchw, B, cfg.in_channel, H, W, g.Hp, g.Wp, x0.p);
        if (r) c.rc = r;
    }
    const std::string enc = "multiscale_encoder";
    int res = cfg.gt_resolution / h->scale;
    Ctx::ConvOpt oin; oin.ksz = 4; oin.pad = 1;
    T t = c.conv(x0, enc + ".in_conv", channels_at(res), oin);
    c.release(x0);
    int bi = 0;
    std::vector<T> outs;              // MultiScaleEncoder.forward returns every block's output (femasr_arch.py:184-192)
    bool t_kept = false;              // t is also held in outs
    for (int i = 0; i < h->encode_depth; ++i, ++bi) {
        const std::string p = enc + ".blocks." + std::to_string(bi);
        Ctx::ConvOpt od; od.stride = 2;
        T d = c.conv(t, p + ".0", channels_at(res / 2), od);
        if (!t_kept) c.release(t);
        t = c.resblock(d, p + ".1", nullptr, true, false, true);
        t = c.resblock(t, p + ".2", nullptr, true);
        res /= 2;
        t_kept = false;
        if (!cfg.lq_stage) {          // HQ stage: enc_feats = outs[::-1]; block i is feats[encode_depth-1-i]
            const int fi = h->encode_depth - 1 - i;
            outs.push_back(t);
            t_kept = fi < h->max_depth && needed(fi) && i + 1 < h->encode_depth;
        }
    }
    std::vector<T> feats(FEMASR_MAX_CODEBOOKS + 1);
    if (cfg.lq_stage) {               // enc_feats = outs[-3:] = (Swin stage, up-block 1, up-block 2)
        t = c.swin_layers(t, enc + ".blocks." + std::to_string(bi++));
        feats[0] = t;
        int last = 0;                 // up-blocks nobody reads are skipped (the reference computes and discards them)
        for (int u = 1; u <= 2 && u < h->max_depth; ++u) if (needed(u)) last = u;
        for (int u = 0; u < last; ++u, ++bi) {
            feats[u + 1] = c.up_block(feats[u], enc + ".blocks." + std::to_string(bi), channels_at(res * 2), nullptr, true);
            res *= 2;
        }
        for (int u = 1; u < last; ++u)
            if (!needed(u)) c.release(feats[u]);
    } else {
        for (int i = 0; i < h->encode_depth; ++i) {
            const int fi = h->encode_depth - 1 - i;
            if (fi < (int)feats.size() && fi < h->max_depth && needed(fi)) feats[fi] = outs[i];
        }
    }
    return run_tail(c, feats[0], feats, fuse_skip, true, idx_out, out_nchw, g.crop_h, g.crop_w);
}

// decode_indices (femasr_arch.py:376-385): codebook 0 -> after_quant_group[0] -> every decoder block -> out_conv
int run_decode_indices(femasr_handle *h, Arena *arena, hipStream_t stream, const int64_t *indices, int B, int hq, int wq,
                       float *out_nchw)
{
    const femasr_config &cfg = h->cfg;
    Ctx c{h, arena, stream};
    T zq = c.alloc_t(B, hq, wq, cfg.e_dim[0]);
    if (!c.rc && !c.dry()) {
        Scope sc(h, c.s(), c.dry(), SLOT_LAYOUT, 0.0, (double)zq.numel() * 8.0);
        const int r = femasr_codebook_gather(c.s(), indices, (int64_t)B * hq * wq, cfg.e_dim[0], c.Wt("quantize_group.0.embedding.weight"), cfg.n_e[0], zq.p);
        if (r) c.rc = r;
    }
    Ctx::ConvOpt oa; oa.lowp = true;
    T x = c.conv(zq, "after_quant_group.0.conv", channels_at(cfg.codebook_scale[0]), oa);
    c.release(zq);
    std::vector<T> feats(FEMASR_MAX_CODEBOOKS + 1);
    return run_tail(c, x, feats, false, false, nullptr, out_nchw, hq << h->max_depth, wq << h->max_depth);
}

int check_ready(const femasr_handle *h)
{
    FEMASR_REQUIRE(h, "null handle");
    if (!h->finalized) return femasr_set_error(FEMASR_ERR_WEIGHT, "weights not finalized: call femasr_set_weight for every key, then femasr_finalize_weights");
    return FEMASR_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char *femasr_last_error(void) { return g_err; }
int femasr_version(void) { return 102; }

int femasr_create(const femasr_config *cfg, femasr_handle **out)
{
    FEMASR_REQUIRE(cfg && out, "create: null argument");
    FEMASR_REQUIRE(cfg->in_channel == 3, "create: in_channel must be 3");
    FEMASR_REQUIRE(cfg->n_codebooks >= 1 && cfg->n_codebooks <= FEMASR_MAX_CODEBOOKS, "create: n_codebooks must be 1..%d", FEMASR_MAX_CODEBOOKS);
    for (int q = 0; q < cfg->n_codebooks; ++q) {
        FEMASR_REQUIRE(cfg->n_e[q] > 0 && cfg->n_e[q] % 128 == 0 && cfg->e_dim[q] > 0 && cfg->e_dim[q] % 32 == 0,
                       "create: codebook %dx%d unsupported (n_e %% 128, e_dim %% 32)", cfg->n_e[q], cfg->e_dim[q]);
        FEMASR_REQUIRE(cfg->codebook_scale[q] > 0 && cfg->gt_resolution % cfg->codebook_scale[q] == 0, "create: bad codebook scale");
        FEMASR_REQUIRE(q == 0 || cfg->codebook_scale[q] > cfg->codebook_scale[q - 1], "create: codebook scales must ascend");
    }
    femasr_handle *h = new femasr_handle();
    h->cfg = *cfg;
    h->scale = cfg->lq_stage ? cfg->scale_factor : 1;
    if (!(h->scale == 1 || h->scale == 2 || h->scale == 4)) { delete h; return femasr_set_error(FEMASR_ERR_INVALID, "create: scale_factor %d unsupported", h->scale); }
    h->max_depth = ilog2(cfg->gt_resolution / cfg->codebook_scale[0]);
    h->last_quant_stage = ilog2(cfg->codebook_scale[cfg->n_codebooks - 1] / cfg->codebook_scale[0]);      // (rows in ascending scale order)
    h->encode_depth = ilog2(cfg->gt_resolution / h->scale / cfg->codebook_scale[0]);
    if (h->max_depth < 1 || h->max_depth > 3) { delete h; return femasr_set_error(FEMASR_ERR_INVALID, "create: max_depth %d unsupported", h->max_depth); }
    for (int q = 1; q < cfg->n_codebooks; ++q) {
        bool hit = false;
        for (int i = 0; i < h->max_depth; ++i) hit = hit || (cfg->gt_resolution / (1 << h->max_depth) * (1 << i)) == cfg->codebook_scale[q];
        if (!hit) { delete h; return femasr_set_error(FEMASR_ERR_INVALID, "create: codebook scale %d is not a decoder resolution", cfg->codebook_scale[q]); }
    }
    DeviceGuard guard(cfg->device);
    if (!guard.ok) { delete h; return femasr_set_error(FEMASR_ERR_HIP, "hipSetDevice(%d) failed", cfg->device); }
    const int rc = build_specs(h);
    if (rc) { delete h; return rc; }
    const int nslots = SLOT_SMALL_COUNT + femasr_conv_variant_count() + femasr_conv_bf16x3_variant_count() + femasr_conv_wino_variant_count() + 1 +
                       femasr_gemm_bf16s_variant_count();
    h->acc_ms.assign(nslots, 0.0); h->acc_flops.assign(nslots, 0.0); h->acc_bytes.assign(nslots, 0.0); h->acc_n.assign(nslots, 0);
    for (int i = 0; i < SLOT_SMALL_COUNT; ++i) h->slot_names.push_back(kSmallNames[i]);
    for (int i = 0; i < femasr_conv_variant_count(); ++i) h->slot_names.push_back(femasr_conv_variant_name(i));
    for (int i = 0; i < femasr_conv_bf16x3_variant_count(); ++i) h->slot_names.push_back(femasr_conv_bf16x3_variant_name(i));
    for (int i = 0; i < femasr_conv_wino_variant_count(); ++i) h->slot_names.push_back(femasr_conv_wino_variant_name(i));
    h->slot_names.push_back(femasr_conv_wino_up2_variant_name());
    for (int i = 0; i < femasr_gemm_bf16s_variant_count(); ++i) h->slot_names.push_back(femasr_gemm_bf16s_variant_name(i));
    *out = h;
    return FEMASR_OK;
}

void femasr_destroy(femasr_handle *h)
{
    if (!h) return;
    for (auto &w : h->specs) { if (w.dev) (void)hipFree(w.dev); if (w.split) (void)hipFree(w.split); if (w.up2w) (void)hipFree(w.up2w); if (w.wino) (void)hipFree(w.wino); if (w.lin3) (void)hipFree(w.lin3); }
    for (int q = 0; q < FEMASR_MAX_CODEBOOKS; ++q) {
        if (h->cbT[q]) (void)hipFree(h->cbT[q]);
        if (h->ee[q]) (void)hipFree(h->ee[q]);
        if (h->vq_aux[q]) (void)hipFree(h->vq_aux[q]);
    }
    for (auto e : h->pool) (void)hipEventDestroy(e);
    for (auto e : h->sub_done) (void)hipEventDestroy(e);
    for (auto st : h->sub_streams) (void)hipStreamDestroy(st);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    delete h;
}

int femasr_num_weights(const femasr_handle *h) { return h ? (int)h->specs.size() : 0; }

int femasr_weight_info(const femasr_handle *h, int i, const char **key, int64_t shape[4], int *ndim)
{
    FEMASR_REQUIRE(h && i >= 0 && i < (int)h->specs.size(), "weight_info: bad index");
    const WSpec &w = h->specs[i];
    if (key) *key = w.key.c_str();
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = k < w.ndim ? w.shape[k] : 1;
    if (ndim) *ndim = w.ndim;
    return FEMASR_OK;
}

int femasr_set_weight(femasr_handle *h, const char *key, const float *dev_ptr, const int64_t *shape, int ndim)
{
    FEMASR_REQUIRE(h && key && dev_ptr && shape, "set_weight: null argument");
    const std::string k(key);
    // buffers that live in checkpoints but are recomputed analytically
    if (k.size() > 24 && (k.rfind(".relative_position_index") == k.size() - 24)) return FEMASR_OK;
    if (k.size() > 10 && (k.rfind(".attn_mask") == k.size() - 10)) return FEMASR_OK;
    auto it = h->index.find(k);
    if (it == h->index.end()) return femasr_set_error(FEMASR_ERR_WEIGHT, "set_weight: unknown key '%s'", key);
    DeviceGuard guard(h->cfg.device);
    FEMASR_REQUIRE(guard.ok, "set_weight: hipSetDevice(%d) failed", h->cfg.device);
    WSpec &w = h->specs[it->second];
    bool same = (ndim == w.ndim);
    for (int i = 0; same && i < ndim; ++i) same = (shape[i] == w.shape[i]);
    if (!same) return femasr_set_error(FEMASR_ERR_WEIGHT, "set_weight: shape mismatch for '%s'", key);
    const size_t n = w.numel();
    size_t alloc = n;
    if (w.kind == W_CONV) alloc = femasr_packed_weight_floats((int)w.shape[0], (int)w.shape[1], (int)w.shape[2], (int)w.shape[3]);
    if (w.kind == W_LINEAR) alloc = femasr_packed_weight_floats((int)w.shape[0], (int)w.shape[1], 1, 1);
    if (!w.dev) FEMASR_CHECK_HIP(hipMalloc((void **)&w.dev, alloc * sizeof(float)));
    int rc = FEMASR_OK;
    if (w.kind == W_CONV)
        rc = femasr_repack_oihw(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], (int)w.shape[2], (int)w.shape[3], w.dev);
    else if (w.kind == W_LINEAR)
        rc = femasr_repack_oihw(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], 1, 1, w.dev);
    else
        FEMASR_CHECK_HIP(hipMemcpyAsync(w.dev, dev_ptr, n * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
    if (rc) return rc;
    // (the encoder's two up-blocks only produce the decoder's skip features, femasr_arch.py:314,361-362: one-codebook networks count them)
    const bool dec_side = behind_every_lookup(h->cfg, h->encode_depth, h->last_quant_stage, k);
    if ((w.kind == W_LINEAR || (w.kind == W_CONV && w.shape[2] == 1 && w.shape[3] == 1)) && (w.shape[1] % 64) == 0) {
        // 1x1 convs / nn.Linear: the three bf16 planes for the fp32-grade product on the bf16 matrix pipe (kernels_gemm_bf16.hip)
        if (!w.lin3) FEMASR_CHECK_HIP(hipMalloc(&w.lin3, femasr_packed_weight_bf16s_bytes((int)w.shape[0], (int)w.shape[1])));
        rc = femasr_repack_k1_bf16s(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], w.lin3);
        if (rc) return rc;
    }
    if (w.kind == W_CONV && !dec_side && !w.up2 && w.shape[2] == 3 && w.shape[3] == 3 && (w.shape[1] % 64) == 0) {
        // a 3x3 conv in FRONT of a codebook lookup (stride 1 or 2): the planes of its (9 Cin x Cout) implicit-GEMM matrix (round 6: linear_math 1
        // runs it as the split-bf16 GEMM)
        if (!w.lin3) FEMASR_CHECK_HIP(hipMalloc(&w.lin3, femasr_packed_weight_conv3x3_bf16s_bytes((int)w.shape[0], (int)w.shape[1])));
        rc = femasr_repack_oihw_bf16s(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], w.lin3);
        if (rc) return rc;
    }
    if (w.kind == W_CONV && w.up2 && w.shape[2] == 3 && w.shape[3] == 3 && (w.shape[1] % 32) == 0) {
        if (!w.up2w) FEMASR_CHECK_HIP(hipMalloc((void **)&w.up2w, femasr_up2_weight_floats((int)w.shape[0], (int)w.shape[1]) * sizeof(float)));
        rc = femasr_repack_oihw_up2(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], w.up2w);
        if (rc) return rc;
    }
    if (w.kind == W_CONV && dec_side && w.shape[2] == 3 && w.shape[3] == 3 && (w.shape[1] % 32) == 0 &&
        (w.shape[0] % 64) == 0) {
        // Winograd-domain weights: F(4x4,3x3) for the plain convs, the 25-component form for the convs behind nn.Upsample(x2)
        const size_t nf = w.up2 ? femasr_wino_up2_weight_floats((int)w.shape[0], (int)w.shape[1]) : femasr_wino_weight_floats((int)w.shape[0], (int)w.shape[1]);
        if (!w.wino) FEMASR_CHECK_HIP(hipMalloc((void **)&w.wino, nf * sizeof(float)));
        rc = w.up2 ? femasr_repack_oihw_wino_up2(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], w.wino)
                   : femasr_repack_oihw_wino(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], w.wino);
        if (rc) return rc;
    }
    if (w.kind == W_CONV && dec_side && w.shape[2] == 3 && w.shape[3] == 3 && (w.shape[1] % 32) == 0) {
        const size_t nb = femasr_packed_weight_bf16x3_bytes((int)w.shape[0], (int)w.shape[1], 3, 3);
        if (!w.split) FEMASR_CHECK_HIP(hipMalloc(&w.split, nb));
        rc = femasr_repack_oihw_bf16x3(nullptr, dev_ptr, (int)w.shape[0], (int)w.shape[1], 3, 3, w.split);
        if (rc) return rc;
    }
    FEMASR_CHECK_HIP(hipStreamSynchronize(nullptr));
    w.set = true;
    h->finalized = false;
    return FEMASR_OK;
}

int femasr_finalize_weights(femasr_handle *h)
{
    FEMASR_REQUIRE(h, "finalize: null handle");
    for (const auto &w : h->specs)
        if (!w.set) return femasr_set_error(FEMASR_ERR_WEIGHT, "finalize: weight '%s' was never set", w.key.c_str());
    DeviceGuard guard(h->cfg.device);
    FEMASR_REQUIRE(guard.ok, "finalize: hipSetDevice(%d) failed", h->cfg.device);
    for (int q = 0; q < h->cfg.n_codebooks; ++q) {
        const int n_e = h->cfg.n_e[q], D = h->cfg.e_dim[q];
        if (!h->cbT[q]) FEMASR_CHECK_HIP(hipMalloc((void **)&h->cbT[q], femasr_packed_weight_floats(n_e, D, 1, 1) * sizeof(float)));
        if (!h->ee[q]) FEMASR_CHECK_HIP(hipMalloc((void **)&h->ee[q], (size_t)n_e * sizeof(float)));
        const float *cb = h->specs[h->index["quantize_group." + std::to_string(q) + ".embedding.weight"]].dev;
        int rc = femasr_repack_oihw(nullptr, cb, n_e, D, 1, 1, h->cbT[q]);     // packed z.e^T operand
        if (rc) return rc;
        rc = femasr_row_sqsum(nullptr, cb, n_e, D, h->ee[q]);
        if (rc) return rc;
        // two-pass exact search (kernels_vq.hip) where the shape allows; FEMASR_VQ=gemm keeps the single-pass fp32 MFMA
        const char *mode = getenv("FEMASR_VQ");
        const bool two = femasr_vq_twopass_ok(n_e, D) && !(mode && !strcmp(mode, "gemm"));
        if (two) {
            if (!h->vq_aux[q]) FEMASR_CHECK_HIP(hipMalloc(&h->vq_aux[q], femasr_vq_aux_bytes(n_e, D)));
            rc = femasr_vq_prepare(nullptr, cb, h->ee[q], n_e, D, h->vq_aux[q]);
            if (rc) return rc;
        } else if (h->vq_aux[q]) {
            (void)hipFree(h->vq_aux[q]);
            h->vq_aux[q] = nullptr;
        }
    }
    FEMASR_CHECK_HIP(hipStreamSynchronize(nullptr));
    h->plans.clear();
    h->finalized = true;
    return FEMASR_OK;
}

// sub-batch i of S over B samples: [lo, hi)
static void sub_range(int B, int S, int i, int *lo, int *hi)
{
    const int q = B / S, r = B % S;
    *lo = i * q + (i < r ? i : r);
    *hi = *lo + q + (i < r ? 1 : 0);
}

// Workspace plan of one call shape: per sub-batch arena peaks (a dry run of the launch schedule), cached per
// (B, H, W, pad_mode) until the stream count / decoder math / weights change.
static int plan_for(femasr_handle *h, int B, int H, int W, int pad_mode, const std::vector<size_t> **out)
{
    const std::array<int, 4> key{B, H, W, pad_mode};
    auto it = h->plans.find(key);
    if (it == h->plans.end()) {
        const int S = std::min(h->nsub, B);
        std::vector<size_t> need;
        for (int i = 0; i < S; ++i) {
            int lo, hi;
            sub_range(B, S, i, &lo, &hi);
            Arena a;
            a.reset(nullptr, 0, true);
            const int rc = run_forward(h, &a, nullptr, nullptr, hi - lo, H, W, pad_mode, nullptr, nullptr);
            if (rc) return rc;
            need.push_back((a.peak + 255) & ~(size_t)255);
        }
        if (h->plans.size() > 64) h->plans.clear();
        it = h->plans.emplace(key, std::move(need)).first;
    }
    *out = &it->second;
    return FEMASR_OK;
}

int femasr_workspace_bytes(const femasr_handle *hc, int B, int H, int W, int pad_mode, size_t *bytes)
{
    femasr_handle *h = const_cast<femasr_handle *>(hc);
    FEMASR_REQUIRE(h && bytes && B > 0 && H > 0 && W > 0, "workspace_bytes: bad args");
    const std::vector<size_t> *need = nullptr;
    const int rc = plan_for(h, B, H, W, pad_mode, &need);
    if (rc) return rc;
    size_t total = 0;
    for (size_t n : *need) total += n;
    *bytes = total + 256;
    return FEMASR_OK;
}

int femasr_forward_shapes(const femasr_handle *h, int H, int W, int pad_mode, int *out_h, int *out_w, int *n_index_maps,
                          int *idx_h, int *idx_w)
{
    FEMASR_REQUIRE(h && H > 0 && W > 0, "forward_shapes: bad args");
    Geometry g;
    const int rc = plan_geometry(h, H, W, pad_mode, &g);
    if (rc) return rc;
    if (out_h) *out_h = g.crop_h;
    if (out_w) *out_w = g.crop_w;
    if (n_index_maps) *n_index_maps = g.nq;
    for (int q = 0; q < g.nq; ++q) {
        if (idx_h) idx_h[q] = g.qh[q];
        if (idx_w) idx_w[q] = g.qw[q];
    }
    return FEMASR_OK;
}

int femasr_set_streams(femasr_handle *h, int n)
{
    FEMASR_REQUIRE(h && n >= 1 && n <= 8, "set_streams: n must be in [1, 8]");
    DeviceGuard guard(h->cfg.device);
    FEMASR_REQUIRE(guard.ok, "set_streams: hipSetDevice(%d) failed", h->cfg.device);
    while ((int)h->sub_streams.size() < n) {
        hipStream_t st;
        hipEvent_t ev;
        FEMASR_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        FEMASR_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->sub_streams.push_back(st);
        h->sub_done.push_back(ev);
    }
    if (!h->ev_fork) FEMASR_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    if (h->nsub != n) h->plans.clear();
    h->nsub = n;
    return FEMASR_OK;
}

static int forward_impl(femasr_handle *h, void *stream, const float *in_nchw, int B, int H, int W, int pad_mode,
                        float *out_nchw, int64_t *indices, void *ws, size_t ws_bytes, int io_u8, int swap_rb)
{
    int rc = check_ready(h);
    if (rc) return rc;
    FEMASR_REQUIRE(in_nchw && out_nchw && ws && B > 0 && H > 0 && W > 0, "forward: bad args");
    FEMASR_REQUIRE(!io_u8 || h->cfg.in_channel == 3, "forward_u8: 3-channel images only");
    FEMASR_REQUIRE(((uintptr_t)ws & 255) == 0, "forward: workspace must be 256-byte aligned");
    DeviceGuard guard(h->cfg.device);
    FEMASR_REQUIRE(guard.ok, "forward: hipSetDevice(%d) failed", h->cfg.device);
    hipStream_t caller = (hipStream_t)stream;
    Geometry g;
    rc = plan_geometry(h, H, W, pad_mode, &g);
    if (rc) return rc;
    // `indices`: the index maps of all codebooks back to back, map q = (B, 1, qh[q], qw[q]) int64
    int64_t *qbase[FEMASR_MAX_CODEBOOKS] = {nullptr, nullptr, nullptr};
    size_t qstride[FEMASR_MAX_CODEBOOKS] = {0, 0, 0};
    {
        size_t off = 0;
        for (int q = 0; q < g.nq; ++q) {
            qstride[q] = (size_t)g.qh[q] * g.qw[q];
            qbase[q] = indices ? indices + off : nullptr;
            off += (size_t)B * qstride[q];
        }
    }
    const std::vector<size_t> *need = nullptr;
    rc = plan_for(h, B, H, W, pad_mode, &need);
    if (rc) return rc;
    const int S = (int)need->size();
    if (S <= 1) {
        Arena a;
        a.reset(ws, ws_bytes, false);
        return run_forward(h, &a, caller, in_nchw, B, H, W, pad_mode, out_nchw, indices ? qbase : nullptr, io_u8, swap_rb);
    }
    // fork: every sub-batch stream waits for the caller's stream; join: the caller's stream waits for all of them.
    // Only event dependencies are added — no host synchronisation.
    size_t total = 0;
    for (size_t n : *need) total += n;
    if (total > ws_bytes) return femasr_set_error(FEMASR_ERR_WORKSPACE, "workspace too small for %d sub-batches", S);
    const size_t in_stride = (size_t)h->cfg.in_channel * H * W * ((io_u8 & 1) ? 1 : 4);            // bytes per sample
    const size_t out_stride = (size_t)3 * g.crop_h * g.crop_w * ((io_u8 & 2) ? 1 : 4);
    FEMASR_CHECK_HIP(hipEventRecord(h->ev_fork, caller));
    size_t off = 0;
    int forked = 0;
    for (int i = 0; i < S && !rc; ++i) {
        int lo, hi;
        sub_range(B, S, i, &lo, &hi);
        Arena a;
        a.reset((char *)ws + off, (*need)[i], false);
        off += (*need)[i];
        hipStream_t st = h->sub_streams[i];
        if (hipStreamWaitEvent(st, h->ev_fork, 0) != hipSuccess) { rc = femasr_set_error(FEMASR_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        forked = i + 1;
        int64_t *qsub[FEMASR_MAX_CODEBOOKS];
        for (int q = 0; q < FEMASR_MAX_CODEBOOKS; ++q) qsub[q] = qbase[q] ? qbase[q] + lo * qstride[q] : nullptr;
        rc = run_forward(h, &a, st, (const float *)((const char *)in_nchw + lo * in_stride), hi - lo, H, W, pad_mode,
                         (float *)((char *)out_nchw + lo * out_stride), indices ? qsub : nullptr, io_u8, swap_rb);
    }
    // join every stream that was forked - also on an error, so that nothing still runs on ws / out when the caller's
    // stream continues
    for (int i = 0; i < forked; ++i) {
        if (hipEventRecord(h->sub_done[i], h->sub_streams[i]) != hipSuccess || hipStreamWaitEvent(caller, h->sub_done[i], 0) != hipSuccess) {
            if (!rc) rc = femasr_set_error(FEMASR_ERR_HIP, "joining sub-batch stream %d failed", i);
        }
    }
    return rc;
}

int femasr_forward(femasr_handle *h, void *stream, const float *in_nchw, int B, int H, int W, int pad_mode,
                   float *out_nchw, int64_t *indices, void *ws, size_t ws_bytes)
{
    return forward_impl(h, stream, in_nchw, B, H, W, pad_mode, out_nchw, indices, ws, ws_bytes, 0, 0);
}

int femasr_forward_u8(femasr_handle *h, void *stream, const uint8_t *in_hwc, int B, int H, int W, int swap_rb, int pad_mode,
                      uint8_t *out_hwc, int64_t *indices, void *ws, size_t ws_bytes)
{
    return forward_impl(h, stream, (const float *)in_hwc, B, H, W, pad_mode, (float *)out_hwc, indices, ws, ws_bytes, 3, swap_rb ? 1 : 0);
}

int femasr_decode_workspace_bytes(const femasr_handle *hc, int B, int hq, int wq, size_t *bytes)
{
    femasr_handle *h = const_cast<femasr_handle *>(hc);
    FEMASR_REQUIRE(h && bytes && B > 0 && hq > 0 && wq > 0, "decode_workspace_bytes: bad args");
    Arena a;
    a.reset(nullptr, 0, true);
    const int rc = run_decode_indices(h, &a, nullptr, nullptr, B, hq, wq, nullptr);
    if (rc) return rc;
    *bytes = a.peak + 256;
    return FEMASR_OK;
}

int femasr_decode_indices(femasr_handle *h, void *stream, const int64_t *indices, int B, int hq, int wq,
                          float *out_nchw, void *ws, size_t ws_bytes)
{
    int rc = check_ready(h);
    if (rc) return rc;
    FEMASR_REQUIRE(indices && out_nchw && ws && B > 0 && hq > 0 && wq > 0, "decode_indices: bad args");
    FEMASR_REQUIRE(((uintptr_t)ws & 255) == 0, "decode_indices: workspace must be 256-byte aligned");
    DeviceGuard guard(h->cfg.device);
    FEMASR_REQUIRE(guard.ok, "decode_indices: hipSetDevice(%d) failed", h->cfg.device);
    Arena a;
    a.reset(ws, ws_bytes, false);
    return run_decode_indices(h, &a, (hipStream_t)stream, indices, B, hq, wq, out_nchw);
}

int femasr_set_decoder_math(femasr_handle *h, int mode)
{
    FEMASR_REQUIRE(h && mode >= 0 && mode <= 3, "set_decoder_math: mode must be 0 (fp32), 1 (bf16x3), 2 (fp32, direct convs only) or 3 (fp32, exact SiLU)");
    if (h->decoder_math != mode) h->plans.clear();
    h->decoder_math = mode;
    return FEMASR_OK;
}

int femasr_debug_set_wino_limits(femasr_handle *h, int log2_total, int log2_image)
{
    FEMASR_REQUIRE(h && log2_total >= 0 && log2_total <= FEMASR_WINO_LOG2_TOTAL && log2_image >= 0 && log2_image <= FEMASR_WINO_LOG2_IMAGE,
                   "debug_set_wino_limits: exponents are 0 (default) or up to %d / %d", FEMASR_WINO_LOG2_TOTAL, FEMASR_WINO_LOG2_IMAGE);
    h->wino_log2_total = log2_total ? log2_total : FEMASR_WINO_LOG2_TOTAL;
    h->wino_log2_image = log2_image ? log2_image : FEMASR_WINO_LOG2_IMAGE;
    h->plans.clear();
    return FEMASR_OK;
}

int femasr_set_linear_math(femasr_handle *h, int mode)
{
    FEMASR_REQUIRE(h && (mode == 0 || mode == 1), "set_linear_math: mode must be 0 (fp32 MFMA chain) or 1 (bf16 three-term split)");
    if (h->linear_math != mode) h->plans.clear();
    h->linear_math = mode;
    return FEMASR_OK;
}

int femasr_profile_enable(femasr_handle *h, int on)
{
    FEMASR_REQUIRE(h, "null handle");
    h->prof = on != 0;
    return FEMASR_OK;
}

static int profile_drain(femasr_handle *h)
{
    for (auto &r : h->recs) {
        FEMASR_CHECK_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        FEMASR_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        h->acc_ms[r.slot] += ms;
        h->acc_flops[r.slot] += r.flops;
        h->acc_bytes[r.slot] += r.bytes;
        h->acc_n[r.slot] += 1;
    }
    h->recs.clear();
    h->pool_used = 0;
    return FEMASR_OK;
}

int femasr_profile_reset(femasr_handle *h)
{
    FEMASR_REQUIRE(h, "null handle");
    const int rc = profile_drain(h);
    std::fill(h->acc_ms.begin(), h->acc_ms.end(), 0.0);
    std::fill(h->acc_flops.begin(), h->acc_flops.end(), 0.0);
    std::fill(h->acc_bytes.begin(), h->acc_bytes.end(), 0.0);
    std::fill(h->acc_n.begin(), h->acc_n.end(), 0);
    return rc;
}

int femasr_profile_slots(const femasr_handle *h) { return h ? (int)h->slot_names.size() : 0; }

const char *femasr_profile_name(const femasr_handle *h, int slot)
{
    return (h && slot >= 0 && slot < (int)h->slot_names.size()) ? h->slot_names[slot].c_str() : "";
}

int femasr_profile_get(femasr_handle *h, int slot, double *ms, int64_t *launches, double *flops, double *bytes)
{
    FEMASR_REQUIRE(h && slot >= 0 && slot < (int)h->slot_names.size(), "profile_get: bad slot");
    const int rc = profile_drain(h);
    if (rc) return rc;
    if (ms) *ms = h->acc_ms[slot];
    if (launches) *launches = h->acc_n[slot];
    if (flops) *flops = h->acc_flops[slot];
    if (bytes) *bytes = h->acc_bytes[slot];
    return FEMASR_OK;
}

}  // extern "C"
