/*
 * femasr_hip.h — flat C ABI of libfemasr_hip.so (MI355X / gfx950 HIP kernels for the
 * FeMaSR SR-inference hot path).
 *
 * The reference is pure PyTorch: it has NO native boundary for this path
 * (SURVEY 2.2).  Each entry point therefore cites the reference *Python* interface
 * whose arithmetic it replaces; `INTEGRATION.md` shows the ctypes binding a
 * reference maintainer would add (it is the binding femasr_amd/_lib.py uses).
 *
 * Conventions
 *   - plain pointers + sizes only; every tensor pointer is a DEVICE pointer owned by the
 *     caller (e.g. torch allocations, `.data_ptr()`); the library never frees or keeps
 *     them past the call, except weights, which `femasr_set_weight` COPIES (repacked)
 *     into allocations the handle owns and frees in `femasr_destroy`.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is
 *     enqueued on it, no hidden synchronisation (except set_weight/finalize, which sync).
 *   - activations inside the library are NHWC fp32 ("tokens (B,HW,C)" == NHWC);
 *     NCHW only at femasr_forward / femasr_decode_indices / pad / crop edges.
 *   - every function returns 0 on success or a negative femasr_status; the message is
 *     available from femasr_last_error() (thread-local).  No C++ exception crosses the ABI.
 *   - a handle is bound to one device and is not thread-safe; one handle per rank.
 */
#ifndef FEMASR_HIP_H
#define FEMASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FEMASR_OK = 0,
    FEMASR_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    FEMASR_ERR_HIP = -2,       /* a HIP runtime call failed                */
    FEMASR_ERR_WEIGHT = -3,    /* unknown key, wrong shape, or missing weight at finalize */
    FEMASR_ERR_WORKSPACE = -4  /* workspace too small                      */
} femasr_status;

typedef struct femasr_handle femasr_handle;

/* Mirrors FeMaSRNet.__init__ keyword arguments (basicsr/archs/femasr_arch.py:216-228).  codebook_params rows
 * [scale, n_e, e_dim] in ascending scale order (femasr_arch.py:231-235); rows after the first are the multi-scale
 * quantisers of femasr_arch.py:277-300 (before_quant on cat(enc_feat, dec_feat), CombineQuantBlock fema_utils.py:87-99). */
#define FEMASR_MAX_CODEBOOKS 3
typedef struct {
    int32_t in_channel;      /* 3 */
    int32_t gt_resolution;   /* 256 */
    int32_t lq_stage;        /* LQ_stage */
    int32_t scale_factor;    /* 4 / 2 (ignored -> 1 when !lq_stage, femasr_arch.py:241) */
    int32_t use_quantize;
    int32_t use_residual;
    int32_t n_codebooks;     /* 1..FEMASR_MAX_CODEBOOKS */
    int32_t codebook_scale[FEMASR_MAX_CODEBOOKS];  /* e.g. {32} */
    int32_t n_e[FEMASR_MAX_CODEBOOKS];             /* e.g. {1024} */
    int32_t e_dim[FEMASR_MAX_CODEBOOKS];           /* e.g. {512} */
    int32_t device;          /* HIP device ordinal the handle is bound to */
} femasr_config;

const char *femasr_last_error(void);
/* 100 * major + minor.  102: the debug hooks moved to femasr_hip_debug.h; femasr_mlp_fused and the process-global femasr_debug_wino_* switches are gone;
 * femasr_extract_tiles_u8 / femasr_paste_tiles_u8 are new (femasr_conv_args is unchanged since 101). */
int femasr_version(void);

/* ---- model handle ------------------------------------------------------------------ */
int femasr_create(const femasr_config *cfg, femasr_handle **out);
void femasr_destroy(femasr_handle *h);

/* Number of weight tensors the configuration expects, and the i-th key / shape
 * (state-dict names and torch layouts of the reference, SURVEY 8b "Weights"). */
int femasr_num_weights(const femasr_handle *h);
int femasr_weight_info(const femasr_handle *h, int i, const char **key, int64_t shape[4], int *ndim);

/* Copy one fp32 tensor (device pointer, torch layout: conv OIHW, linear (out,in), vectors)
 * into the handle, repacking conv/linear weights to the K-major layout of femasr_conv_args.w.
 * Replaces nn.Module.load_state_dict for the path (inference_femasr.py:40). */
int femasr_set_weight(femasr_handle *h, const char *key, const float *dev_ptr,
                      const int64_t *shape, int ndim);
/* Verify every expected weight was set; build derived tensors (codebook^T, |e|^2). */
int femasr_finalize_weights(femasr_handle *h);

/* Number of internal streams the batch is split over (default 1).  Samples are independent (every op is
 * per-sample), so sub-batches run concurrently on separate streams, forked from / joined to the caller's
 * stream by events only; results are bit-identical for any n.  Call before femasr_workspace_bytes. */
int femasr_set_streams(femasr_handle *h, int n);

/* pad_mode 1 = FeMaSRNet.test geometry (mirror-pad to (h/wsz+1)*wsz, crop to h*s; femasr_arch.py:449-468)
 * pad_mode 0 = FeMaSRNet.forward (no pad, output (H*s_out) as produced; femasr_arch.py:470-479)          */
int femasr_workspace_bytes(const femasr_handle *h, int B, int H, int W, int pad_mode, size_t *bytes);

/* Shapes of one call: the output image (out_h, out_w), and per codebook the index map (idx_h[q], idx_w[q]) (arrays of
 * FEMASR_MAX_CODEBOOKS ints).  Follows the reference geometry exactly, including test()'s truncated mirror pad of images
 * smaller than the pad when there is no Swin stage (femasr_arch.py:454-465). */
int femasr_forward_shapes(const femasr_handle *h, int H, int W, int pad_mode, int *out_h, int *out_w, int *n_index_maps,
                          int *idx_h, int *idx_w);

/* Whole hot path: in NCHW fp32 (B,3,H,W) -> out NCHW fp32 (B,3,out_h,out_w) and, if non-NULL, the VQ index maps int64
 * of all codebooks back to back, map q = (B,1,idx_h[q],idx_w[q]).  Replaces FeMaSRNet.encode_and_decode
 * (femasr_arch.py:311-374) inside .test / .forward.  `ws` must be >= femasr_workspace_bytes and 256-byte aligned.
 * Every entry point runs on the handle's device and restores the caller's current device before returning. */
int femasr_forward(femasr_handle *h, void *stream, const float *in_nchw, int B, int H, int W,
                   int pad_mode, float *out_nchw, int64_t *indices, void *ws, size_t ws_bytes);

/* The same forward on uint8 images, the step either side of the path fused into its first and last kernel: in (B,H,W,3)
 * uint8 HWC (swap_rb = 1: cv2's BGR) -> (float)u8 / 255.0f inside the mirror-pad kernel (img2tensor + `/255.`: basicsr/utils/img_util.py:9-35,
 * inference_femasr.py:54-56); out (B,out_h,out_w,3) uint8 HWC = clamp [0,1], x255, round half to even inside the crop kernel (tensor2img,
 * img_util.py:38-94, inference_femasr.py:64-67).  No fp32 NCHW image makes a pass of its own; same bits as femasr_image_u8_to_f32 ->
 * femasr_forward -> femasr_image_f32_to_u8. */
int femasr_forward_u8(femasr_handle *h, void *stream, const uint8_t *in_hwc, int B, int H, int W, int swap_rb, int pad_mode,
                      uint8_t *out_hwc, int64_t *indices, void *ws, size_t ws_bytes);
int femasr_pad_u8hwc_to_nhwc(void *stream, const uint8_t *in, int B, int H, int W, int swap_rb, int Hp, int Wp, float *out);
int femasr_crop_nhwc_to_u8hwc(void *stream, const float *in, int B, int Hs, int Ws, int Hc, int Wc, int swap_rb, uint8_t *out);

/* indices (B,1,h,w) int64 -> image NCHW (B,3,8h,8w).  Replaces FeMaSRNet.decode_indices
 * (femasr_arch.py:376-385) incl. VectorQuantizer.get_codebook_entry (:102-112). */
int femasr_decode_workspace_bytes(const femasr_handle *h, int B, int hq, int wq, size_t *bytes);
int femasr_decode_indices(femasr_handle *h, void *stream, const int64_t *indices, int B, int hq, int wq,
                          float *out_nchw, void *ws, size_t ws_bytes);

/* Per-kernel timing (HIP events recorded on `stream` around every launch) while enabled.
 * Slots = one per conv_igemm instantiation + one per small-kernel family; femasr_profile_get
 * synchronises the recorded events and returns, for launches since the last reset, the summed
 * milliseconds, launch count and algorithmic FLOPs (0 for the HBM-bound families) / bytes. */
int femasr_profile_enable(femasr_handle *h, int on);
int femasr_profile_reset(femasr_handle *h);
int femasr_profile_slots(const femasr_handle *h);
const char *femasr_profile_name(const femasr_handle *h, int slot);
int femasr_profile_get(femasr_handle *h, int slot, double *ms, int64_t *launches, double *flops, double *bytes);

/* ---- per-kernel entry points (unit parity tests; same kernels femasr_forward launches) ---- */

/* test() pad + layout: NCHW (B,C,H,W) -> NHWC (B,Hp,Wp,C), mirror rows/cols (femasr_arch.py:454-460). */
int femasr_pad_nchw_to_nhwc(void *stream, const float *in, int B, int C, int H, int W, int Hp, int Wp, float *out);
/* crop + layout: NHWC (B,Hs,Ws,C) -> NCHW (B,C,Hc,Wc) top-left (femasr_arch.py:464-465). */
int femasr_crop_nhwc_to_nchw(void *stream, const float *in, int B, int Hs, int Ws, int C, int Hc, int Wc, float *out);

enum { FEMASR_PRO_NONE = 0, FEMASR_PRO_GN_SILU = 1, FEMASR_PRO_LN = 2 /* retired: LayerNorm is femasr_layernorm, a separate pass */ };
enum { FEMASR_ACT_NONE = 0, FEMASR_ACT_GELU = 1 };

/* Convolution / linear on fp32 MFMA.  Replaces nn.Conv2d (femasr_arch.py:150,159,173,
 * 203,273,298; fema_utils.py:75,78,90; network_swinir.py:465), nn.Upsample(x2) fused on load
 * (femasr_arch.py:172,202), nn.Linear (network_swinir.py:19-21,105-112; ksz=1 on (1,rows,1,Cin)),
 * GroupNorm-apply+SiLU fused on load, bias / GELU / residual adds fused on store.
 * Kernel families: 3x3 stride-1 pad-1 with Cin % 32 == 0 -> halo kernels; 1x1 stride-1 with Cin % 32 == 0 (every
 * nn.Linear) -> the LDS-DMA GEMM; everything else -> the general implicit GEMM. */
typedef struct {
    const float *in;      /* (B,H,W,Cin) NHWC, pre-upsample size */
    int32_t B, H, W, Cin;
    const float *w;       /* packed weights as emitted by femasr_repack_oihw (fragment-major, see there) */
    const float *bias;    /* [Cout] */
    int32_t Cout, ksz, stride, pad, up2;
    int32_t prologue;     /* FEMASR_PRO_* */
    const float *pro_a;   /* GN: a[B][Cin] */
    const float *pro_b;   /* GN: b[B][Cin] */
    const float *pro_c;   /* unused (was LN beta) */
    int32_t act;          /* FEMASR_ACT_* (applied after bias, before residuals) */
    const float *res1;    /* optional (B,Ho,Wo,Cout) added after act   */
    const float *res2;    /* optional second residual                   */
    float *out;           /* (B,Ho,Wo,Cout) */
    int32_t Ho, Wo;
    const void *w_bf16x3; /* optional: femasr_repack_oihw_bf16x3 weights.  When non-NULL and the layer is a 3x3
                             stride-1 pad-1 conv with Cin % 32 == 0 (no LN prologue / GELU) it runs on the bf16
                             matrix cores with the 3-term hi/lo split (~1e-5 relative, NOT bit-exact); NULL =
                             exact fp32 */
    double *gn_part;      /* optional, 3x3 stride-1 halo convs (exact fp32 and bf16x3): per-(sample, 8x16 tile, group)
                             partial GroupNorm(32) moments (sum, sum of squares) of the OUTPUT,
                             [B][tilesY*tilesX][32][2] doubles, for femasr_gn_coeffs_from_partials (saves the separate
                             moments pass over the tensor).  fp32 path: exactly the partials of the specified
                             summation order (bit-identical coefficients to femasr_gn_coeffs); needs Cout/32 a power of
                             two <= 32 (bf16x3: <= 8); anything else is refused.  With up2 on the exact path (phase filters) the
                             partials are per HALF-resolution tile and phase: [B][4*ceil(H/8)*ceil(W/16)][32][2], index
                             tile*4 + 2a + b (oracle: orc_gn_coeffs(..., phases = 1)). */
    const float *w_up2;   /* up2 = 1 on a 3x3 stride-1 pad-1 conv with Cin % 32 == 0 (exact fp32 path): the four phase
                             matrices of femasr_repack_oihw_up2 -- nn.Upsample(x2, nearest) + Conv2d evaluated as four
                             2x2-tap filters on the low-resolution input (REQUIRED for that shape; other up2 shapes
                             read the upsampled image through `w`). */
    const float *w_wino;  /* optional: femasr_repack_oihw_wino weights.  When non-NULL the layer (3x3 stride-1 pad-1, no
                             x2, Cin % 32 == 0, Cout % 64 == 0, no GELU) runs in the Winograd F(4x4,3x3) form: fp32
                             throughout, 4x fewer multiplies, results within fp32 rounding (~1e-5 relative at worst) of the
                             direct form and bit-identical to oracle/femasr_oracle.c orc_conv3x3_winograd.  The model uses
                             it for the convs behind the codebook lookup only (they cannot move a VQ index).  With gn_part the
                             partials are per 16x16-pixel sub-block: [B][ceil(H/16)*ceil(W/16)][32][2] (orc_gn_coeffs mode 2).
                             With up2 = 1: femasr_repack_oihw_wino_up2 weights, the 25-product form of nearest-x2 + conv
                             (no prologue; partials per 16x16-pixel OUTPUT sub-block; orc_conv_up2_winograd). */
    int32_t fast_act;     /* Winograd convs with the GN+SiLU prologue only: 1 = SiLU through the hardware exp2 / rcp units
                             (v_exp_f32, v_rcp_f32: 1 ulp each) instead of the IEEE-exact polynomial + division - ~6x fewer
                             VALU instructions in the staging; output within ~1e-6 relative of the exact form (no longer
                             bit-identical to the oracle).  0 = exact. */
    const float *in_add;  /* optional second INPUT tensor of the same (B,H,W,Cin) shape: the conv reads in + in_add (one fp32 add per
                             element while staging).  Only the x2 Winograd-type form takes it (up2 = 1 with w_wino; anything else refuses):
                             FeMaSRNet's decoder adds the encoder's skip feature to a stage's input (`x = x + enc_feats[i]`,
                             femasr_arch.py:361-362) right in front of that stage's x2 conv, so the sum never makes a pass of its own. */
    const void *w_bf16s;  /* optional (struct version 101): femasr_repack_k1_bf16s weights.  When non-NULL the layer - a 1x1 stride-1 conv /
                             nn.Linear with Cin % 64 == 0, no prologue (network_swinir.py:19-21,105-107,121,143; femasr_arch.py:298), or (round 6, weights from
                             femasr_repack_oihw_bf16s) a 3x3 stride-1 pad-1 conv with Cin % 64 == 0, no prologue, no activation - runs on
                             the bf16 matrix pipe as an fp32-GRADE product: both operands split exactly into three bf16 terms, the six
                             partial products of relative size >= 2^-16 accumulated in fp32 (two accumulators), ~3x closer to the fp64
                             result than the fp32 fmaf chain and bit-identical to oracle/femasr_oracle.c orc_linear_bf16s, which restates
                             the instruction's accumulation arithmetic from hardware probes (kernels_gemm_bf16.hip).  `w` is not read. */
} femasr_conv_args;
int femasr_conv2d(void *stream, const femasr_conv_args *a);

/* GroupNorm(32,eps) moments folded into per-(n,c) scale/shift: y = fmaf(x,a,b) (fema_utils.py:22).
 * scratch: >= femasr_gn_scratch_bytes(B,H,W,C,G) bytes. */
size_t femasr_gn_scratch_bytes(int B, int H, int W, int C, int G);
int femasr_gn_coeffs(void *stream, const float *x, int B, int H, int W, int C, int G,
                     const float *gamma, const float *beta, float eps, float *a, float *b, void *scratch);
/* Same as femasr_gn_coeffs, from the partial moments a halo conv wrote with femasr_conv_args.gn_part
 * (tiles = ceil(H/8)*ceil(W/16) of the producing conv's output). */
int femasr_gn_coeffs_from_partials(void *stream, const double *part, int B, int tiles, int H, int W, int C, int G,
                                   const float *gamma, const float *beta, float eps, float *a, float *b);
/* y = silu(fmaf(x, a[n][c], b[n][c])) on an NHWC tensor: the GroupNorm-apply + SiLU of a ResBlock conv (fema_utils.py:73-74,76-77) as a
 * pass of its own, in the arithmetic of the conv kernels' GN+SiLU prologue (IEEE-exact SiLU; oracle: orc_scale_shift_silu).  Used in
 * front of the 3x3 convs that run as the split-bf16 GEMM (w_bf16s with ksz = 3), which take their input without a prologue. */
int femasr_gn_silu_apply(void *stream, const float *x, int B, int H, int W, int C, const float *a, const float *b, float *y);
/* LayerNorm(C=256) row moments -> stats[rows][2] = (mean, rstd) (network_swinir.py:199,205). */
int femasr_ln_stats(void *stream, const float *x, int64_t rows, int C, float eps, float *stats);
/* y = LayerNorm(x) over the last dim (C = 256): the same moments, then fmaf((x-mean)*rstd, gamma, beta)
 * (network_swinir.py:243,277: norm1 / norm2 ahead of qkv / fc1). */
int femasr_layernorm(void *stream, const float *x, int64_t rows, int C, const float *gamma, const float *beta,
                     float eps, float *y);
/* 8x8 (shifted-)window multi-head attention incl. rel-pos bias and shift mask
 * (network_swinir.py:114-145, 216-237, 249-272).  qkv (B,H*W,3C) -> out (B,H*W,C), natural token order. */
int femasr_window_attention(void *stream, const float *qkv, int B, int H, int W, int C, int heads, int shift,
                            const float *table, float *out);
/* VectorQuantizer.forward (femasr_arch.py:35-38,50-100): z (M,D) rows; cbT = femasr_repack_oihw(cb, n_e, D, 1, 1);
 * ee[j] = |e_j|^2 (femasr_row_sqsum).  Writes idx (M) int64 first-min, zq (M,D) straight-through.
 * scratch: >= (M*(n_e/128)*2 + M + 64) floats. */
int femasr_vq(void *stream, const float *z, int64_t M, int D, const float *cb, const float *cbT,
              const float *ee, int n_e, int64_t *idx, float *zq, void *scratch);
int femasr_row_sqsum(void *stream, const float *x, int64_t rows, int D, float *out);
/* The same lookup as a two-pass EXACT search (kernels_vq.hip): a bf16-MFMA pass keeps, from a rigorous error bound, the
 * few codes per row that can be the fp32 first-min; the specified fp32 chain is then evaluated for those only.
 * Results are bit-identical to femasr_vq.  Shapes: e_dim a power of two in 64..512, n_e % 64 == 0, n_e <= 1024
 * (femasr_vq_twopass_ok); aux = femasr_vq_prepare(cb, ee) image of femasr_vq_aux_bytes bytes;
 * scratch >= femasr_vq_scratch_bytes(M, n_e) bytes (covers femasr_vq too). */
int femasr_vq_twopass_ok(int n_e, int D);
size_t femasr_vq_aux_bytes(int n_e, int D);
int femasr_vq_prepare(void *stream, const float *cb, const float *ee, int n_e, int D, void *aux);
size_t femasr_vq_scratch_bytes(int64_t M, int n_e);
int femasr_vq_twopass(void *stream, const float *z, int64_t M, int D, const float *cb, const void *aux, const float *ee,
                      int n_e, int64_t *idx, float *zq, void *scratch);
/* Pass 1 alone: cand (M,32) uint16 candidate codes, cnt (M) uint16 count (0xFFFF = every code). */
int femasr_vq_candidates(void *stream, const float *z, int64_t M, int D, const void *aux, const float *ee, int n_e,
                         uint16_t *cand, uint16_t *cnt);
int femasr_codebook_gather(void *stream, const int64_t *idx, int64_t M, int D, const float *cb, int n_e, float *zq);
/* out (B,H,W,Ca+Cb) = cat(a (B,H,W,Ca), nearest-resize(b (B,Hb,Wb,Cb) -> H x W)) along channels: torch.cat((x, y), 1) of
 * femasr_arch.py:334 (Hb,Wb == H,W) and CombineQuantBlock's F.interpolate + cat (fema_utils.py:92-99; source index
 * floor(dst * Hb / H), exact for the integer ratios the architecture produces). */
int femasr_concat_resize(void *stream, const float *a, int Ca, const float *b, int Hb, int Wb, int Cb, int B, int H, int W,
                         float *out);
/* test_tile (femasr_arch.py:387-447) on the device.  extract: the n input windows of ONE shape class (th x tw, top-left corners
 * yx_dev[2k], yx_dev[2k+1], int32 on the device) of a (B,C,H,W) image -> a (n*B, C, th, tw) batch (tile-major, as
 * torch.cat of the reference's crops).  paste: the upscaled tile bodies -> the (B,C,Ho,Wo) canvas;
 * rects_dev[6k..] = (src_y, src_x, dst_y, dst_x, h, w) in upscaled pixels, hmax = max h.  Tiles of different ranks are pasted
 * after the all-gather, which stays in the host's torch.distributed (RCCL) group: see INTEGRATION.md. */
int femasr_extract_tiles(void *stream, const float *in, int B, int C, int H, int W, const int32_t *yx_dev, int n, int th, int tw,
                         float *out);
int femasr_paste_tiles(void *stream, const float *tiles, int B, int C, int n, int th, int tw, const int32_t *rects_dev, int hmax,
                       int Ho, int Wo, float *out);
/* The same on uint8 HWC images (3 channels): crops of a (B,H,W,3) image -> (n*B, th, tw, 3), ready for femasr_forward_u8; tile bodies of
 * (n*B, th, tw, 3) -> the (B,Ho,Wo,3) canvas.  FeMaSRNet.test_tile_u8 and the all-gather of the multi-GPU path move these bytes - a
 * quarter of the fp32 tiles' - and the CLI's tiled branch (inference_femasr.py:58-67) never holds an fp32 image. */
int femasr_extract_tiles_u8(void *stream, const uint8_t *in, int B, int H, int W, const int32_t *yx_dev, int n, int th, int tw, uint8_t *out);
int femasr_paste_tiles_u8(void *stream, const uint8_t *tiles, int B, int n, int th, int tw, const int32_t *rects_dev, int hmax,
                          int Ho, int Wo, uint8_t *out);

/* OIHW -> the packed FRAGMENT-MAJOR weight layout of femasr_conv_args.w (also nn.Linear (out,in) with kh=kw=1
 * and the codebook for femasr_vq):  out[q][ntile][lane][kk], zero padded, with
 *   K index k = ((ci/32)*kh*kw + ky*kw + kx)*32 + ci%32 when I % 32 == 0, else (ky*kw + kx)*I + ci;
 *   q = k/32, kk = (k%32)/2, lane = (k&1)*32 + o%32, ntile = o/32.
 * 1x1 layers with I % 32 == 0 (kh = kw = 1: nn.Linear, 1x1 convs, the codebook) get the GEMM layout instead:
 *   out[q][ntile][j][lane][t] = W[o = 32*ntile + lane%32][k = 32q + 8j + 4*(lane/32) + t]   (same size).
 * 3x3 layers with O <= 4 and I % 32 == 0 (out_conv) additionally carry a compact [k][4] copy behind the matrix (the
 * direct VALU kernel reads it through scalar loads); femasr_packed_weight_floats includes it.
 * `out` must hold femasr_packed_weight_floats(O,I,kh,kw) floats. */
size_t femasr_packed_weight_floats(int O, int I, int kh, int kw);
int femasr_repack_oihw(void *stream, const float *in, int O, int I, int kh, int kw, float *out);
/* 3x3 OIHW -> femasr_conv_args.w_up2: phase (a,b) of nearest-x2 + conv reads input rows {y-1+a, y+a} and columns
 * {x-1+b, x+b}; slot weights are the fp32 sums of the taps that land on the same input pixel (rows summed over kx first,
 * ascending; then over ky, ascending).  out: femasr_up2_weight_floats(O, I) floats = 4 x femasr_packed_weight_floats(O,I,2,2). */
size_t femasr_up2_weight_floats(int O, int I);
/* 3x3 OIHW -> femasr_conv_args.w_wino: U = G g G^T (6x6 per (o, i)) of F(4x4,3x3), down the columns then along the rows in
 * the operation order of oracle/femasr_oracle.c orc_g6, stored [Cin/8][36 components][Cout/32][lane][4] (the MFMA B fragments
 * of one 8-channel step: 1 KiB per wave load).  out: femasr_wino_weight_floats(O, I) floats = 36 * I * 32*ceil(O/32).
 * The layout is private to the library (pack and launch through it). */
size_t femasr_wino_weight_floats(int O, int I);
int femasr_repack_oihw_wino(void *stream, const float *in, int O, int I, float *out);
/* The same for the conv behind nn.Upsample(x2, nearest) (femasr_arch.py:172-173,202-203): a 4x4 output tile reads a 4x4 patch of
 * the low-resolution input, which leaves 5 products per dimension - 25 components per (o, i) instead of 36, layout
 * [Cin/8][25][Cout/32][lane][4].  Pass the result as femasr_conv_args.w_wino together with up2 = 1 (no prologue; Cin % 32 == 0,
 * Cout % 64 == 0).  GroupNorm partial moments (gn_part) come per 16x16-pixel OUTPUT sub-block, as for the F(4x4,3x3) form. */
size_t femasr_wino_up2_weight_floats(int O, int I);
int femasr_repack_oihw_wino_up2(void *stream, const float *w_oihw, int O, int I, float *out);
int femasr_repack_oihw_up2(void *stream, const float *in, int O, int I, float *out);

/* (out, in) fp32 -> femasr_conv_args.w_bf16s: the three bf16 planes of every weight (w = w1 + w2 + w3 exactly), packed
 * [Cin/16][ceil(Cout/32)][plane][lane][8 bf16] - the MFMA B fragments of one 16-channel step, 1 KiB per (column tile, plane). */
size_t femasr_packed_weight_bf16s_bytes(int O, int I);
int femasr_repack_k1_bf16s(void *stream, const float *w_oi, int O, int I, void *out);
/* The same for a 3x3 conv weight (O, I, 3, 3), I % 64 == 0: the planes of the (9 I x O) matrix of the conv's implicit GEMM, k = (3 ky + kx) I + c.
 * With such weights in w_bf16s a 3x3 stride-1 pad-1 conv (no prologue, no activation; bias and residual operands as usual) runs as the
 * split-bf16 GEMM over K = 9 Cin: the same arithmetic as the linear layers (oracle: conv3x3_bf16s = im2col + orc_linear_bf16s).  FeMaSRNet
 * (linear_math 'bf16_split') uses it for the 3x3 convs in FRONT of the codebook lookup - the encoder's ResBlocks (femasr_arch.py:150-164,
 * fema_utils.py:65-84) and the conv behind every RSTB (network_swinir.py:465). */
size_t femasr_packed_weight_conv3x3_bf16s_bytes(int O, int I);
int femasr_repack_oihw_bf16s(void *stream, const float *w_oihw, int O, int I, void *out);
/* Arithmetic of the network's 1x1 convs / nn.Linear layers (the Swin qkv / proj / fc1 / fc2 and before_quant):
 * 1 (default, 'bf16_split'): the fp32-grade product on the bf16 matrix pipe described at femasr_conv_args.w_bf16s;
 * 0 ('fp32'): one fp32 fmaf chain per output on the fp32 MFMA (kernels_gemm.hip), bit-identical to OracleNet(linear_math='fp32'). */
int femasr_set_linear_math(femasr_handle *h, int mode);

/* Split-bf16 (hi/lo) fragment-major weights for the opt-in bf16x3 conv path (3x3 convs behind the VQ lookup). */
size_t femasr_packed_weight_bf16x3_bytes(int O, int I, int kh, int kw);
int femasr_repack_oihw_bf16x3(void *stream, const float *in, int O, int I, int kh, int kw, void *out);
/* 0 (default, 'fp32'): every layer fp32; in single-codebook networks the 3x3 convs that do NOT feed the codebook lookup
 *    (after_quant, DecoderBlocks, the LQ encoder's two up-blocks) run in the Winograd F(4x4,3x3) form (4x fewer multiplies)
 *    with the GroupNorm+SiLU prologue on the hardware exp2 / rcp units (femasr_conv_args.fast_act): same fp32 accuracy against
 *    the reference (~1e-5), VQ indices bit-exact (everything that feeds a lookup stays in the exact direct form), the image
 *    within ~1e-5 of the oracle.
 * 3 ('fp32_strict'): the same with the IEEE-exact SiLU: bit-identical to the oracle (OracleNet()).
 * 1: those convs (and the x2 convs) use the bf16x3 path instead: output within the north-star 1e-3 bound of the fp32 result
 *    (measured ~1e-4).
 * 2 ('fp32_direct'): fp32 with every conv in the direct form, bit-identical to OracleNet(winograd=False). */
int femasr_set_decoder_math(femasr_handle *h, int mode);

/* ---- image pre / post-processing (the steps either side of the path) ---- */
/* uint8 HWC (3 channels; swap_rb=1 for cv2-style BGR input) -> fp32 CHW RGB in [0,1]: (float)u8 / 255.0f.
 * Replaces img2tensor(...)/255. (basicsr/utils/img_util.py:9-35, inference_femasr.py:54-56). */
int femasr_image_u8_to_f32(void *stream, const uint8_t *in_hwc, int H, int W, int swap_rb, float *out_chw);
/* fp32 CHW RGB -> clamp [0,1] -> (x*255).round() half-to-even -> uint8 HWC (swap_rb=1: BGR for cv2.imwrite).
 * Replaces tensor2img (basicsr/utils/img_util.py:38-94). */
int femasr_image_f32_to_u8(void *stream, const float *in_chw, int H, int W, int swap_rb, uint8_t *out_hwc);

/* ---- measurement support ---- */
/* Sustained-clock probe: FEMASR_CLOCK_PROBE_BLOCKS blocks of 4 waves stream `mfmas_per_wave` back-to-back fp32 MFMAs (the load
 * of the hot kernels; 64 shader cycles each) and write the s_memtime ticks of their loop to ticks[blocks*4].  ticks / wall time
 * of the launch = the clock the chip sustains under MFMA load at that moment (bench.py reports it before and after the timed
 * region: the network's kernels are clock-bound, so a throttling box shows up here and not as a kernel regression). */
#define FEMASR_CLOCK_PROBE_BLOCKS 256
int femasr_clock_probe(void *stream, int mfmas_per_wave, unsigned long long *ticks);
int femasr_clock_probe_entries(void);      /* number of 64-bit entries femasr_clock_probe writes (size the buffer from this) */
/* Test / tuning hooks (process-global block-shape overrides, the per-handle Winograd size limit, the MFMA probe) are declared in
 * include/femasr_hip_debug.h: they are exported by the library but are not part of the drop-in interface. */
#ifdef __cplusplus
}
#endif
#endif /* FEMASR_HIP_H */
