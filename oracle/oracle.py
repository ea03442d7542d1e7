"""Python driver of the CPU oracle (oracle/femasr_oracle.c).

>>> TEST INFRASTRUCTURE ONLY — see the header of femasr_oracle.c. <<<
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

`OracleNet` orchestrates the C ops into the reference's forward
(basicsr/archs/femasr_arch.py:311-374 encode_and_decode, :449-468 test,
:387-447 test_tile, :376-385 decode_indices) from a plain {key: ndarray}
state dict with the reference's key names and OIHW / (out,in) weight layouts.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libfemasr_oracle.so')
_SRC = os.path.join(_HERE, 'femasr_oracle.c')
_CFLAGS = ['-O3', '-mavx2', '-mfma', '-ffp-contract=off', '-fno-math-errno', '-fopenmp',
           '-shared', '-fPIC', '-std=gnu11']


def build(force=False):
    """Compile the C restatement (gcc).  No -march=native: the .so travels to the GPU box."""
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    subprocess.check_call(['gcc'] + _CFLAGS + ['-o', _SO, _SRC, '-lm'])
    return _SO


_lib = None
_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        # one thread per PHYSICAL core unless the caller chose (on SMT hosts - this dev container, the GPU box - the default of one
        # thread per logical CPU runs the integer-heavy loops slower).  Set through the OpenMP runtime the oracle's .so is linked
        # against, NOT through os.environ: the environment variable would also change torch's CPU thread pool if torch initialises
        # afterwards.
        if 'OMP_NUM_THREADS' not in os.environ:
            try:
                ctypes.CDLL('libgomp.so.1').omp_set_num_threads(max(1, (os.cpu_count() or 2) // 2))
            except OSError:
                pass
        i, i64, f, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
        L.orc_math_eval.argtypes = [i, vp, vp, i64]
        L.orc_pad_nchw_to_nhwc.argtypes = [vp, i, i, i, i, i, i, vp]
        L.orc_crop_nhwc_to_nchw.argtypes = [vp, i, i, i, i, i, i, vp]
        L.orc_conv2d.argtypes = [vp, i, i, i, i, vp, vp, i, i, i, i, i, i, vp, vp, vp, i, i]
        L.orc_gn_coeffs.argtypes = [vp, i, i, i, i, i, vp, vp, f, vp, vp, i]
        L.orc_scale_shift_silu.argtypes = [vp, i, i64, i, vp, vp, vp]
        L.orc_layernorm.argtypes = [vp, i64, i, vp, vp, f, vp]
        L.orc_window_attention.argtypes = [vp, i, i, i, i, i, i, vp, vp]
        L.orc_vq.argtypes = [vp, i64, i, vp, i, vp, vp, vp, vp]
        L.orc_codebook_gather.argtypes = [vp, i64, i, vp, vp]
        L.orc_conv3x3_winograd.argtypes = [vp, i, i, i, i, vp, vp, i, vp, vp, vp]
        L.orc_conv_up2_winograd.argtypes = [vp, i, i, i, i, vp, vp, i, vp, vp, vp]
        L.orc_linear_bf16s.argtypes = [vp, i64, i, vp, vp, i, i, vp, vp, vp, i]
        L.orc_mfma_dot8_v8.argtypes = [vp, vp, vp, vp]
        L.orc_split3_eval.argtypes = [vp, i64, vp, vp, vp]
        L.orc_mfma_dot8.argtypes = [f, vp, vp]
        for fn in ('orc_math_eval', 'orc_pad_nchw_to_nhwc', 'orc_crop_nhwc_to_nchw', 'orc_conv2d', 'orc_conv3x3_winograd', 'orc_conv_up2_winograd',
                   'orc_gn_coeffs', 'orc_scale_shift_silu', 'orc_layernorm', 'orc_window_attention',
                   'orc_vq', 'orc_codebook_gather', 'orc_linear_bf16s', 'orc_split3_eval', 'orc_mfma_dot8_v8'):
            getattr(L, fn).restype = None
        L.orc_mfma_dot8.restype = f
        _lib = L
    return _lib


def _c(a, dtype=np.float32):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------ op wrappers
def math_eval(which, x):
    x = _c(x)
    y = np.empty_like(x)
    lib().orc_math_eval({'exp': 0, 'erf': 1, 'silu': 2, 'gelu': 3}[which], _p(x), _p(y), x.size)
    return y


def pad_nchw_to_nhwc(x, hp, wp):
    x = _c(x)
    b, c, h, w = x.shape
    out = np.empty((b, hp, wp, c), np.float32)
    lib().orc_pad_nchw_to_nhwc(_p(x), b, c, h, w, hp, wp, _p(out))
    return out


def crop_nhwc_to_nchw(x, hc, wc):
    x = _c(x)
    b, hs, ws, c = x.shape
    out = np.empty((b, c, hc, wc), np.float32)
    lib().orc_crop_nhwc_to_nchw(_p(x), b, hs, ws, c, hc, wc, _p(out))
    return out


def repack_conv_weight(w_oihw):
    """OIHW (torch Conv2d) -> [kh][kw][Cin][Cout]."""
    return _c(np.transpose(np.asarray(w_oihw), (2, 3, 1, 0)))


def repack_linear_weight(w_oi):
    """(out,in) (torch Linear) -> [1][1][in][out]."""
    return _c(np.asarray(w_oi).T)


def winograd_ok(cin, cout, ksz, stride, pad, up2, act=0):
    """The shapes the kernels run in the Winograd F(4x4,3x3) form when the caller marks the conv as behind the codebook
    lookup (femasr_conv_args.w_wino): 3x3 stride-1 pad-1, no x2, Cin % 32 == 0, Cout % 64 == 0."""
    return ksz == 3 and stride == 1 and pad == 1 and not up2 and act == 0 and cin % 32 == 0 and cout % 64 == 0


WINO_LOG2_LIMITS = [31, 27]      # total / per-image element limits; tests move both sides together (femasr_debug_wino_limits)


def wino_fits(b, h, w, cin, cout, up2=False):
    """The kernels' size limits (32-bit buffer offsets: femasr_conv_wino_shape_ok / femasr_conv_wino_up2_shape_ok) are part of the
    rule: a larger layer runs in the direct / phase-filter form on the GPU, so it does here.  (b, h, w) = the conv's INPUT."""
    tot, img = 2 ** WINO_LOG2_LIMITS[0], 2 ** WINO_LOG2_LIMITS[1]
    if not (b * h * w * cin < tot and h * w * cin < img and cin <= 1024):
        return False
    if up2:
        return b * 4 * h * w * cout < tot and 4 * h * w * cout < img and 25 * cin * cout < 2 ** 29
    return b * h * w * cout < tot and h * w * cout < img and 36 * cin * cout < 2 ** 29


def winograd_up2_ok(cin, cout, ksz, stride, pad, up2, act=0):
    """nn.Upsample(x2) + 3x3 conv in the 25-product Winograd-type form (orc_conv_up2_winograd; kernels_wino_up2.hip)."""
    return ksz == 3 and stride == 1 and pad == 1 and bool(up2) and act == 0 and cin % 32 == 0 and cout % 64 == 0


def conv2d(x, w_khwc, bias, ksz, stride=1, pad=0, up2=False, act=0, res1=None, res2=None, wino=False):
    x = _c(x)
    b, h, w, cin = x.shape
    cout = w_khwc.shape[-1]
    wino = wino and wino_fits(b, h, w, cin, cout, up2)
    if wino and winograd_up2_ok(cin, cout, ksz, stride, pad, up2, act):
        out = np.empty((b, 2 * h, 2 * w, cout), np.float32)
        res1 = None if res1 is None else _c(res1)
        res2 = None if res2 is None else _c(res2)
        lib().orc_conv_up2_winograd(_p(x), b, h, w, cin, _p(_c(w_khwc)), _p(_c(bias)), cout, _p(res1), _p(res2), _p(out))
        return out
    if wino and winograd_ok(cin, cout, ksz, stride, pad, up2, act):
        out = np.empty((b, h, w, cout), np.float32)
        res1 = None if res1 is None else _c(res1)
        res2 = None if res2 is None else _c(res2)
        lib().orc_conv3x3_winograd(_p(x), b, h, w, cin, _p(_c(w_khwc)), _p(_c(bias)), cout, _p(res1), _p(res2), _p(out))
        return out
    hv, wv = (2 * h, 2 * w) if up2 else (h, w)
    ho = (hv + 2 * pad - ksz) // stride + 1
    wo = (wv + 2 * pad - ksz) // stride + 1
    out = np.empty((b, ho, wo, cout), np.float32)
    w_khwc = _c(w_khwc)
    bias = _c(bias)
    if res1 is not None:
        res1 = _c(res1)
        assert res1.size == out.size
    if res2 is not None:
        res2 = _c(res2)
        assert res2.size == out.size
    lib().orc_conv2d(_p(x), b, h, w, cin, _p(w_khwc), _p(bias), cout, ksz, stride, pad, int(up2), act,
                     _p(res1), _p(res2), _p(out), ho, wo)
    return out


def split_ok(cin, ksz=1, stride=1, pad=0, up2=False):
    """The 1x1 / Linear shapes csrc/kernels_gemm_bf16.hip takes in linear_math 'bf16_split' (femasr_gemm_bf16s_shape_ok)."""
    return ksz == 1 and stride == 1 and pad == 0 and not up2 and cin % 64 == 0


def mfma_dot8(d, a_bits, b_bits):
    """One 8-product group of v_mfma_f32_*_bf16 (orc_mfma_dot8): a, b = 8 bf16 bit patterns (uint16), d = the accumulator."""
    a = np.ascontiguousarray(a_bits, np.uint16)
    b = np.ascontiguousarray(b_bits, np.uint16)
    assert a.size == 8 and b.size == 8
    return np.float32(lib().orc_mfma_dot8(np.float32(d), _p(a), _p(b)))


def split3(x):
    """x = x1 + x2 + x3, three bf16 values (orc_split3): returns the three uint16 bit-pattern arrays."""
    x = _c(x)
    p = [np.empty(x.shape, np.uint16) for _ in range(3)]
    lib().orc_split3_eval(_p(x), x.size, _p(p[0]), _p(p[1]), _p(p[2]))
    return p


def mfma_dot8_v8(d8, a_bits, b_bits_kmajor):
    """The 8-column AVX2 form of mfma_dot8 (orc_dot8_v8): d8 (8,), a (8,) shared by the lanes, b (8 k, 8 lanes)."""
    d = _c(d8).reshape(8)
    a = np.ascontiguousarray(a_bits, np.uint16).reshape(8)
    b = np.ascontiguousarray(b_bits_kmajor, np.uint16).reshape(8, 8)
    out = np.empty(8, np.float32)
    lib().orc_mfma_dot8_v8(_p(d), _p(a), _p(b), _p(out))
    return out


def linear_bf16s(x_rows, w_oi, bias, act=0, res1=None, res2=None, scalar=False):
    """The 1x1 / Linear layers in the split-bf16 arithmetic (orc_linear_bf16s): x (rows, Cin), w_oi (Cout, Cin) = torch's layout."""
    x = _c(x_rows)
    rows, cin = x.shape
    w = _c(w_oi)
    cout = w.shape[0]
    assert w.shape[1] == cin and cin % 16 == 0
    out = np.empty((rows, cout), np.float32)
    res1 = None if res1 is None else _c(res1).reshape(rows, cout)
    res2 = None if res2 is None else _c(res2).reshape(rows, cout)
    lib().orc_linear_bf16s(_p(x), rows, cin, _p(w), _p(_c(bias)), cout, act, _p(res1), _p(res2), _p(out), int(scalar))
    return out


def conv3x3_split_ok(cin, ksz, stride, pad, up2, numel=0):
    """3x3 pad-1 convs of stride 1 or 2 the library runs as the split-bf16 GEMM over K = 9 Cin (femasr_conv_args.w_bf16s with ksz = 3;
    csrc/kernels_gemm_bf16.hip CONV form) - round 6: the convs that FEED the codebook lookup, in linear_math 'bf16_split'."""
    return ksz == 3 and stride in (1, 2) and pad == 1 and not up2 and cin % 64 == 0 and cin <= 1024 and numel < 2 ** 31      # (femasr_conv3x3_bf16s_shape_ok)


def conv3x3_bf16s(x, w_khwc, bias, res1=None, res2=None, stride=1):
    """3x3 stride-1 pad-1 conv (femasr_arch.py:150-164, fema_utils.py:75,78, network_swinir.py:465) in the split-bf16 arithmetic: the
    implicit GEMM out[pixel][o] = sum_k A[pixel][k] W[k][o] with k = (3 ky + kx) Cin + c - tap-major, channels ascending inside a tap,
    zeros outside the image - evaluated EXACTLY as orc_linear_bf16s evaluates a K = 9 Cin linear layer (16-deep steps, six partial
    products, two accumulators, the instruction's restated arithmetic).  x (B,H,W,Cin) NHWC, w_khwc [3][3][Cin][Cout]."""
    x = _c(x)
    b, h, w, cin = x.shape
    cout = w_khwc.shape[-1]
    xp = np.zeros((b, h + 2, w + 2, cin), np.float32)
    xp[:, 1:-1, 1:-1] = x
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1            # (h + 2 - 3) // stride + 1
    cols = np.empty((b, ho, wo, 9, cin), np.float32)
    for ky in range(3):
        for kx in range(3):
            cols[:, :, :, 3 * ky + kx] = xp[:, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride]
    w_oi = _c(np.asarray(w_khwc, np.float32).reshape(9 * cin, cout).T)
    rows = b * ho * wo
    y = linear_bf16s(cols.reshape(rows, 9 * cin), w_oi, bias, 0,
                     None if res1 is None else _c(res1).reshape(rows, cout), None if res2 is None else _c(res2).reshape(rows, cout))
    return y.reshape(b, ho, wo, cout)


def linear(x_tokens, w_io, bias, act=0, res=None, split=False):
    """x: (..., Cin) -> (..., Cout); w_io is [in][out].  split: the bf16-pipe arithmetic (linear_bf16s) where the shape allows."""
    shp = x_tokens.shape
    rows = int(np.prod(shp[:-1]))
    if split and split_ok(shp[-1]):
        y = linear_bf16s(_c(x_tokens).reshape(rows, shp[-1]), _c(np.asarray(w_io).reshape(shp[-1], -1).T), bias, act, res)
        return y.reshape(shp[:-1] + (y.shape[-1],))
    y = conv2d(_c(x_tokens).reshape(1, rows, 1, shp[-1]), w_io, bias, 1, act=act,
               res1=None if res is None else _c(res).reshape(1, rows, 1, -1))
    return y.reshape(shp[:-1] + (w_io.shape[-1],))


def gn_fusable(c):
    """Channel counts whose GroupNorm(32) partial moments the 3x3 kernels emit from their epilogue (csrc/common.h)."""
    cg = c // 32
    return c % 32 == 0 and 1 <= cg <= 32 and (cg & (cg - 1)) == 0


def gn_coeffs(x, gamma, beta, eps=1e-6, groups=32, phases=False):
    """phases: x is the output of a conv whose epilogue produced the partial moments in its own tile order -- only the ORDER of
    the fp64 partial sums differs.  True / 1: a phase-filter x2 conv (per half-resolution tile and phase); 2: a Winograd conv
    (per 16x16-pixel sub-block, orc_gn_wino_partial)."""
    x = _c(x)
    b, h, w, c = x.shape
    a = np.empty((b, c), np.float32)
    bb = np.empty((b, c), np.float32)
    gamma, beta = _c(gamma), _c(beta)
    lib().orc_gn_coeffs(_p(x), b, h, w, c, groups, _p(gamma), _p(beta), eps, _p(a), _p(bb), int(phases))
    return a, bb


def scale_shift_silu(x, a, b):
    x = _c(x)
    bsz, h, w, c = x.shape
    y = np.empty_like(x)
    a, b = _c(a), _c(b)
    lib().orc_scale_shift_silu(_p(x), bsz, h * w, c, _p(a), _p(b), _p(y))
    return y


def gn_silu(x, gamma, beta, phases=False):
    a, b = gn_coeffs(x, gamma, beta, phases=phases)
    return scale_shift_silu(x, a, b)


def layernorm(x, gamma, beta, eps=1e-5):
    x = _c(x)
    c = x.shape[-1]
    rows = x.size // c
    y = np.empty_like(x)
    gamma, beta = _c(gamma), _c(beta)
    lib().orc_layernorm(_p(x), rows, c, _p(gamma), _p(beta), eps, _p(y))
    return y


def window_attention(qkv, b, h, w, c, heads, shift, table):
    qkv = _c(qkv)
    out = np.empty((b, h * w, c), np.float32)
    table = _c(table)
    lib().orc_window_attention(_p(qkv), b, h, w, c, heads, shift, _p(table), _p(out))
    return out


def vq(z_rows, codebook, want_dists=False):
    z_rows = _c(z_rows)
    m, d = z_rows.shape
    codebook = _c(codebook)
    idx = np.empty((m,), np.int64)
    zq = np.empty_like(z_rows)
    dmin = np.empty((m,), np.float32) if want_dists else None
    d2 = np.empty((m,), np.float32) if want_dists else None
    lib().orc_vq(_p(z_rows), m, d, _p(codebook), codebook.shape[0], _p(idx), _p(zq), _p(dmin), _p(d2))
    return (idx, zq, dmin, d2) if want_dists else (idx, zq)


def codebook_gather(idx, codebook):
    idx = np.ascontiguousarray(idx, np.int64).reshape(-1)
    codebook = _c(codebook)
    zq = np.empty((idx.size, codebook.shape[1]), np.float32)
    lib().orc_codebook_gather(_p(idx), idx.size, codebook.shape[1], _p(codebook), _p(zq))
    return zq


# ------------------------------------------------------------------ image pre / post (host glue of the reference)
def image_u8_to_f32(img_hwc_u8, bgr=False):
    """img2tensor(...)/255. (basicsr/utils/img_util.py:9-35; inference_femasr.py:55): uint8 HWC -> fp32 (1,3,H,W) RGB."""
    img = np.asarray(img_hwc_u8)
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    if bgr:
        img = img[:, :, ::-1]
    return (img.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0))[None]


def image_f32_to_u8(x_nchw, bgr=False):
    """tensor2img (img_util.py:38-94): clamp [0,1], CHW->HWC, optional RGB->BGR, (x*255).round() (half to even), uint8."""
    x = np.clip(np.asarray(x_nchw, np.float32)[0], 0.0, 1.0).transpose(1, 2, 0)
    if bgr:
        x = x[:, :, ::-1]
    return np.round(x * np.float32(255.0)).astype(np.uint8)


# ------------------------------------------------------------------ network
_CHANNELS = {8: 256, 16: 256, 32: 256, 64: 256, 128: 128, 256: 64, 512: 32}   # femasr_arch.py:244-252


class OracleNet:
    """Reference forward restated on the C ops.  `sd` = {key: ndarray} with the reference key names."""

    def __init__(self, sd, codebook_params=((32, 1024, 512),), gt_resolution=256, LQ_stage=False,
                 scale_factor=4, use_quantize=True, use_residual=True, winograd=True, linear_math='bf16_split'):
        self.sd = {k: np.asarray(v) for k, v in sd.items()}
        # arithmetic of the 1x1 / Linear layers: 'bf16_split' (the kernels' default since round 5: kernels_gemm_bf16.hip) or
        # 'fp32' (one fmaf chain per output on the fp32 MFMA: kernels_gemm.hip)
        assert linear_math in ('fp32', 'bf16_split')
        self.lin_split = linear_math == 'bf16_split'
        self.conv_split = self.lin_split          # the 3x3 convs that feed the lookup follow the same switch (csrc/model.hip Ctx::conv)
        # the kernels' default exact-fp32 mode: 3x3 convs behind the codebook lookup of a single-codebook network run in
        # the Winograd F(4x4,3x3) form (model.hip Ctx::conv `wino`); winograd=False = decoder_math 'fp32_direct'
        self.wino = bool(winograd)
        self.single = len(codebook_params) == 1
        self.cb_scales = [int(c[0]) for c in codebook_params]
        self.LQ_stage = bool(LQ_stage)
        self.scale_factor = int(scale_factor) if LQ_stage else 1
        self.gt_res = int(gt_resolution)
        self.codebook_scale = int(codebook_params[0][0])
        self.use_quantize = use_quantize
        self.use_residual = use_residual
        self.max_depth = int(math.log2(self.gt_res // self.codebook_scale))
        # decoder stage of the LAST lookup: with several codebooks only after_quant_group[last], decoder_group[i >= that stage] and out_conv
        # are behind every lookup (model.hip behind_every_lookup); with one codebook the whole decoder side and the LQ up-blocks are
        self.last_quant_stage = int(math.log2(max(self.cb_scales) // self.codebook_scale))
        self.encode_depth = int(math.log2(self.gt_res // self.scale_factor // self.codebook_scale))
        self.probes = None
        self._wcache = {}

    # -- weight access (repacked once)
    def _conv_w(self, prefix):
        if prefix not in self._wcache:
            self._wcache[prefix] = (repack_conv_weight(self.sd[prefix + '.weight']), _c(self.sd[prefix + '.bias']))
        return self._wcache[prefix]

    def _lin_w(self, prefix):
        if prefix not in self._wcache:
            self._wcache[prefix] = (repack_linear_weight(self.sd[prefix + '.weight']), _c(self.sd[prefix + '.bias']))
        return self._wcache[prefix]

    def _probe(self, name, val):
        if self.probes is not None:
            self.probes[name] = val

    # -- blocks
    def _conv(self, x, prefix, ksz, stride=1, pad=1, up2=False, res1=None, res2=None, dec=False):
        """dec: the conv sits behind the codebook lookup (decoder side)."""
        w, b = self._conv_w(prefix)
        if self.conv_split and not dec and conv3x3_split_ok(x.shape[-1], ksz, stride, pad, up2, x.size):
            # round 6: a 3x3 conv in front of the codebook lookup (encoder ResBlocks, RSTB tail convs) as the split-bf16 GEMM over K = 9 Cin
            return conv3x3_bf16s(x, w, b, res1, res2, stride)
        return conv2d(x, w, b, ksz, stride, pad, up2, 0, res1, res2, wino=dec and self.wino)

    def _wino_fused(self, c, dec, shape=None):
        """The conv that produced a c-channel tensor ran in the Winograd form AND emitted the GroupNorm partials (model.hip
        Ctx::conv: wino_on && gn_ok): the next GroupNorm sums them in the sub-block order (orc_gn_coeffs mode 2)."""
        fits = shape is None or wino_fits(shape[0], shape[1], shape[2], c, c)
        return dec and self.wino and fits and winograd_ok(c, c, 3, 1, 1, False) and gn_fusable(c)

    def _resblock(self, x, prefix, res2=None, dec=False, from_up2=None, from_wino=False):
        # fema_utils.py:65-84: conv2(silu(gn2(conv1(silu(gn1(x)))))) + x   (+ optional fused skip add)
        # from_up2 = Cin of the x2 conv that produced x: its phase-filter kernel emitted the moments (phase-tile order);
        # from_wino: x is the output of the previous ResBlock's Winograd conv, which emitted the moments (sub-block order)
        c = x.shape[-1]
        ph = 0
        if from_up2 is not None and gn_fusable(c):
            if dec and self.wino and winograd_up2_ok(from_up2, c, 3, 1, 1, True) and wino_fits(x.shape[0], x.shape[1] // 2, x.shape[2] // 2, from_up2, c, True):
                ph = 2          # the x2 conv ran in the 25-product Winograd-type form: moments per 16x16 output sub-block
            elif from_up2 % 32 == 0:
                ph = 1          # phase-filter form: moments per half-resolution tile and phase
        elif from_wino and self._wino_fused(c, dec, x.shape):
            ph = 2
        t = gn_silu(x, self.sd[prefix + '.conv.0.norm.weight'], self.sd[prefix + '.conv.0.norm.bias'], phases=ph)
        t = self._conv(t, prefix + '.conv.2', 3, dec=dec)
        t = gn_silu(t, self.sd[prefix + '.conv.3.norm.weight'], self.sd[prefix + '.conv.3.norm.bias'],
                    phases=2 if self._wino_fused(c, dec, x.shape) else 0)
        return self._conv(t, prefix + '.conv.5', 3, res1=x, res2=res2, dec=dec)

    def _swin_block(self, x, b, h, w, prefix, shift):
        # network_swinir.py:239-279
        c = x.shape[-1]
        t = layernorm(x, self.sd[prefix + '.norm1.weight'], self.sd[prefix + '.norm1.bias'])
        wq, bq = self._lin_w(prefix + '.attn.qkv')
        qkv = linear(t, wq, bq, split=self.lin_split)
        att = window_attention(qkv, b, h, w, c, 8, shift, self.sd[prefix + '.attn.relative_position_bias_table'])
        wp, bp = self._lin_w(prefix + '.attn.proj')
        x = linear(att, wp, bp, res=x, split=self.lin_split)
        t = layernorm(x, self.sd[prefix + '.norm2.weight'], self.sd[prefix + '.norm2.bias'])
        w1, b1 = self._lin_w(prefix + '.mlp.fc1')
        hdn = linear(t, w1, b1, act=1, split=self.lin_split)
        w2, b2 = self._lin_w(prefix + '.mlp.fc2')
        return linear(hdn, w2, b2, res=x, split=self.lin_split)

    def _swin_layers(self, x, prefix):
        # femasr_arch.py:114-132 + network_swinir.py:442-482 (tokens (B,HW,C) == NHWC)
        b, h, w, c = x.shape
        for r in range(4):
            rp = f'{prefix}.swin_blks.{r}'
            y = x.reshape(b, h * w, c)
            for k in range(6):
                y = self._swin_block(y, b, h, w, f'{rp}.residual_group.blocks.{k}', 0 if k % 2 == 0 else 4)
            x = self._conv(y.reshape(b, h, w, c), rp + '.conv', 3, res1=x)
            self._probe(f'swin_rstb{r}', x)
        return x

    def _encoder(self, x):
        p = 'multiscale_encoder'
        x = self._conv(x, p + '.in_conv', 4, 1, 1)
        self._probe('in_conv', x)
        outs = []
        bi = 0
        for _ in range(self.encode_depth):
            x = self._conv(x, f'{p}.blocks.{bi}.0', 3, 2, 1)
            x = self._resblock(x, f'{p}.blocks.{bi}.1')
            x = self._resblock(x, f'{p}.blocks.{bi}.2')
            self._probe(f'enc_block{bi}', x)
            outs.append(x)
            bi += 1
        if self.LQ_stage:
            x = self._swin_layers(x, f'{p}.blocks.{bi}')
            self._probe(f'enc_block{bi}', x)
            outs.append(x)
            bi += 1
            for _ in range(2):
                cin = x.shape[-1]
                x = self._conv(x, f'{p}.blocks.{bi}.1', 3, 1, 1, up2=True, dec=self.single)
                x = self._resblock(x, f'{p}.blocks.{bi}.2', dec=self.single, from_up2=cin)       # the LQ up-blocks only make the decoder's skip features
                x = self._resblock(x, f'{p}.blocks.{bi}.3', dec=self.single, from_wino=True)
                self._probe(f'enc_block{bi}', x)
                outs.append(x)
                bi += 1
        return outs

    def _decoder_block(self, x, i, res2=None):
        p = f'decoder_group.{i}.block'
        cin = x.shape[-1]
        dec = self.single or i >= self.last_quant_stage
        x = self._conv(x, p + '.1', 3, 1, 1, up2=True, dec=dec)
        x = self._resblock(x, p + '.2', dec=dec, from_up2=cin)
        return self._resblock(x, p + '.3', res2=res2, dec=dec, from_wino=True)

    def encode_and_decode(self, x_nhwc):
        """femasr_arch.py:311-374; returns (out NHWC, [indices (B,1,h,w) int64 per codebook])."""
        feats = self._encoder(x_nhwc)
        feats = feats[-3:] if self.LQ_stage else feats[::-1]
        fuse_skip = self.LQ_stage and self.use_residual
        indices = []
        prev_dec, prev_q, qi = None, None, 0
        x = feats[0]
        for i in range(self.max_depth):
            cur_res = self.gt_res // 2 ** self.max_depth * 2 ** i
            if cur_res in self.cb_scales:        # quantise at this scale (femasr_arch.py:332-359)
                zin = feats[i] if prev_dec is None else np.concatenate((feats[i], prev_dec), axis=-1)
                if self.lin_split and split_ok(zin.shape[-1]):      # 1x1 conv == Linear over the pixels
                    wq_, bq_ = self._conv_w(f'before_quant_group.{qi}')
                    z = linear(zin, wq_.reshape(zin.shape[-1], -1), bq_, split=True)
                else:
                    z = self._conv(zin, f'before_quant_group.{qi}', 1, 1, 0)
                if qi == 0:
                    self._probe('z', z)
                b, h, w, d = z.shape
                idx, zq = vq(z.reshape(-1, d), self.sd[f'quantize_group.{qi}.embedding.weight'])
                zq = zq.reshape(b, h, w, d)
                if qi == 0:
                    self._probe('z_q', zq)
                indices.append(idx.reshape(b, 1, h, w))
                if not self.use_quantize:
                    zq = z
                ain = zq
                if prev_q is not None:           # CombineQuantBlock: nearest resize of the previous scale + concat (fema_utils.py:92-99)
                    ys = (np.arange(h) * prev_q.shape[1]) // h
                    xs = (np.arange(w) * prev_q.shape[2]) // w
                    ain = np.concatenate((zq, prev_q[:, ys][:, :, xs]), axis=-1)
                x = self._conv(ain, f'after_quant_group.{qi}.conv', 3, dec=self.single or qi == len(self.cb_scales) - 1)
                if qi == 0:
                    self._probe('after_quant', x)
                prev_q = zq
                qi += 1
            elif fuse_skip:
                # `x = x + enc_feats[i]` (femasr_arch.py:361-362): the kernels fold this add into the previous block's last
                # conv epilogue (`+ res2`, after `+ res1`); the same fp32 additions in the same order
                pass
            nxt_skip = (fuse_skip and i + 1 < self.max_depth and
                        (self.gt_res // 2 ** self.max_depth * 2 ** (i + 1)) not in self.cb_scales)
            x = self._decoder_block(x, i, res2=feats[i + 1] if nxt_skip else None)
            self._probe(f'dec{i}' + ('_plus_skip' if nxt_skip else ''), x)
            prev_dec = x
        wout, bout = self._conv_w('out_conv')
        out = conv2d(x, wout, bout, 3, 1, 1)
        return out, indices

    # -- public surface (NCHW in / out like the reference module)
    def forward(self, x_nchw):
        x = _c(x_nchw)
        b, c, h, w = x.shape
        out, idx = self.encode_and_decode(pad_nchw_to_nhwc(x, h, w))
        y = crop_nhwc_to_nchw(out, out.shape[1], out.shape[2])
        return (y, idx[0]) if len(idx) == 1 else (y, idx)

    def test(self, x_nchw, return_indices=False):
        x = _c(x_nchw)
        b, c, h, w = x.shape
        wsz = 8 // self.scale_factor * 8
        hp = (h // wsz + 1) * wsz
        wp = (w // wsz + 1) * wsz
        if not self.LQ_stage:        # torch.cat([x, flip(x)])[..., :h+pad] holds at most 2h rows (femasr_arch.py:459-460)
            hp, wp = min(hp, 2 * h), min(wp, 2 * w)
        out, idx = self.encode_and_decode(pad_nchw_to_nhwc(x, hp, wp))
        y = crop_nhwc_to_nchw(out, min(h * self.scale_factor, out.shape[1]), min(w * self.scale_factor, out.shape[2]))
        if not return_indices:
            return y
        return (y, idx[0]) if len(idx) == 1 else (y, idx)

    def test_tile(self, x_nchw, tile_size=240, tile_pad=16):
        x = _c(x_nchw)
        b, c, h, w = x.shape
        s = self.scale_factor
        out = np.zeros((b, c, h * s, w * s), np.float32)
        for ty in range(math.ceil(h / tile_size)):
            for tx in range(math.ceil(w / tile_size)):
                x0, y0 = tx * tile_size, ty * tile_size
                x1, y1 = min(x0 + tile_size, w), min(y0 + tile_size, h)
                x0p, y0p = max(x0 - tile_pad, 0), max(y0 - tile_pad, 0)
                x1p, y1p = min(x1 + tile_pad, w), min(y1 + tile_pad, h)
                t = self.test(x[:, :, y0p:y1p, x0p:x1p])
                oy, ox = (y0 - y0p) * s, (x0 - x0p) * s
                out[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = t[:, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]
        return out

    def decode_indices(self, indices):
        indices = np.asarray(indices)
        assert indices.ndim == 4
        b, _, h, w = indices.shape
        cb = self.sd['quantize_group.0.embedding.weight']
        zq = codebook_gather(indices, cb).reshape(b, h, w, -1)
        x = self._conv(zq, 'after_quant_group.0.conv', 3, dec=self.single or len(self.cb_scales) == 1)
        for i in range(self.max_depth):
            x = self._decoder_block(x, i)
        wout, bout = self._conv_w('out_conv')
        out = conv2d(x, wout, bout, 3, 1, 1)
        return crop_nhwc_to_nchw(out, out.shape[1], out.shape[2])
