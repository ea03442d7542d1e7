"""FeMaSRNet on MI355X: the reference's module surface over libfemasr_hip.so.

Drop-in for basicsr/archs/femasr_arch.py:214-479 on the INFERENCE path:
same class name, registered in ARCH_REGISTRY the same way, same keyword-only
constructor (femasr_arch.py:216-228), same methods (`test`, `test_tile`,
`forward`, `encode_and_decode`, `decode_indices`), same public attributes and
the same state-dict keys/shapes, so `load_state_dict(torch.load(p)['params'],
strict=False)` (inference_femasr.py:40) works on released checkpoints.

The torch sub-modules below are PARAMETER HOLDERS only: they give the state
dict its names, shapes and default initialisation.  No torch op computes
anything on the path — every forward goes through the C ABI
(include/femasr_hip.h) into the HIP kernels, and raises if the extension or a
GPU is missing (no CPU / eager fallback).
"""
import ctypes
import math

import numpy as np
import torch
from torch import nn

from .. import _lib
from .. import tiling
from ..registry import ARCH_REGISTRY

_CHANNELS = {8: 256, 16: 256, 32: 256, 64: 256, 128: 128, 256: 64, 512: 32}   # femasr_arch.py:244-252


# --------------------------------------------------------------------------- parameter holders
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - never used on the product path
        raise RuntimeError('parameter holder: computation happens in libfemasr_hip.so')


def _seq(*mods):
    """Numbered children like nn.Sequential, without its forward."""
    h = _Holder()
    for i, m in enumerate(mods):
        if m is not None:
            h.add_module(str(i), m)
    return h


def _norm(c):               # fema_utils.py:5-29 NormLayer('gn') -> .norm = GroupNorm(32, c, eps=1e-6)
    h = _Holder()
    h.norm = nn.GroupNorm(32, c, eps=1e-6, affine=True)
    return h


def _resblock(c):           # fema_utils.py:65-84: conv = Sequential(norm, act, conv, norm, act, conv)
    h = _Holder()
    h.conv = _seq(_norm(c), None, nn.Conv2d(c, c, 3, 1, 1), _norm(c), None, nn.Conv2d(c, c, 3, 1, 1))
    return h


def _swin_block(dim, heads, ws, shift, res=(32, 32)):   # network_swinir.py:164-237
    h = _Holder()
    h.norm1 = nn.LayerNorm(dim)
    attn = _Holder()
    attn.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), heads))
    nn.init.trunc_normal_(attn.relative_position_bias_table, std=.02)
    ys, xs = np.divmod(np.arange(ws * ws), ws)
    rel = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
    attn.register_buffer('relative_position_index', torch.from_numpy(rel.astype(np.int64)))
    attn.qkv = nn.Linear(dim, dim * 3)
    attn.proj = nn.Linear(dim, dim)
    h.attn = attn
    h.norm2 = nn.LayerNorm(dim)
    mlp = _Holder()
    mlp.fc1 = nn.Linear(dim, dim * 4)
    mlp.fc2 = nn.Linear(dim * 4, dim)
    h.mlp = mlp
    if shift > 0:           # persistent buffer in checkpoints (16x64x64 for the (32,32) construction size)
        hh, ww = res
        lab = np.zeros((hh, ww), np.float32)
        sl = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
        cnt = 0
        for a in sl:
            for b in sl:
                lab[a, b] = cnt
                cnt += 1
        win = lab.reshape(hh // ws, ws, ww // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
        diff = win[:, None, :] - win[:, :, None]
        h.register_buffer('attn_mask', torch.from_numpy(np.where(diff != 0, -100.0, 0.0).astype(np.float32)))
    else:
        h.register_buffer('attn_mask', None)
    return h


def _rstb(dim=256, depth=6, heads=8, ws=8):             # network_swinir.py:419-482
    h = _Holder()
    grp = _Holder()
    grp.blocks = _seq(*[_swin_block(dim, heads, ws, 0 if i % 2 == 0 else ws // 2) for i in range(depth)])
    h.residual_group = grp
    h.conv = nn.Conv2d(dim, dim, 3, 1, 1)
    return h


def _up_block(cin, cout):   # Upsample, Conv, ResBlock, ResBlock  (femasr_arch.py:171-176, 201-206)
    return _seq(None, nn.Conv2d(cin, cout, 3, 1, 1), _resblock(cout), _resblock(cout))


# --------------------------------------------------------------------------- the arch
@ARCH_REGISTRY.register()
class FeMaSRNet(nn.Module):
    def __init__(self, *, in_channel=3, codebook_params=None, gt_resolution=256, LQ_stage=False,
                 norm_type='gn', act_type='silu', use_quantize=True, scale_factor=4,
                 use_semantic_loss=False, use_residual=True, **ignore_kwargs):
        super().__init__()
        codebook_params = np.array(codebook_params)
        if codebook_params.ndim != 2 or codebook_params.shape[1] != 3 or not 1 <= codebook_params.shape[0] <= _lib.MAX_CODEBOOKS:
            raise ValueError(f'codebook_params must be rows of [scale, n_e, e_dim] (1..{_lib.MAX_CODEBOOKS} rows)')
        if norm_type != 'gn' or act_type != 'silu':
            raise NotImplementedError("MI355X path builds norm_type='gn', act_type='silu' (the published configs)")
        self.codebook_scale = codebook_params[:, 0]
        codebook_emb_num = codebook_params[:, 1].astype(int)
        codebook_emb_dim = codebook_params[:, 2].astype(int)
        self.use_quantize = use_quantize
        self.in_channel = in_channel
        self.gt_res = gt_resolution
        self.LQ_stage = LQ_stage
        self.scale_factor = scale_factor if LQ_stage else 1
        self.use_residual = use_residual
        # accepted for signature compatibility; the VGG branch is training-only and is forced off
        # inside test() by the reference as well (femasr_arch.py:451-452)
        self.use_semantic_loss = False
        self.max_depth = int(np.log2(gt_resolution // self.codebook_scale[0]))
        encode_depth = int(np.log2(gt_resolution // self.scale_factor // self.codebook_scale[0]))
        self._encode_depth = encode_depth

        # ---- parameter tree (names == reference state-dict keys)
        enc = _Holder()
        res = gt_resolution // self.scale_factor
        enc.in_conv = nn.Conv2d(in_channel, _CHANNELS[res], 4, padding=1)
        blocks = []
        for _ in range(encode_depth):
            ci, co = _CHANNELS[res], _CHANNELS[res // 2]
            blocks.append(_seq(nn.Conv2d(ci, co, 3, stride=2, padding=1), _resblock(co), _resblock(co)))
            res //= 2
        if LQ_stage:
            swin = _Holder()
            swin.swin_blks = _seq(*[_rstb() for _ in range(4)])
            blocks.append(swin)
            for _ in range(2):
                blocks.append(_up_block(_CHANNELS[res], _CHANNELS[res * 2]))
                res *= 2
        enc.blocks = _seq(*blocks)
        self.multiscale_encoder = enc

        dec = []
        out_ch = None
        for i in range(self.max_depth):
            r = gt_resolution // 2 ** self.max_depth * 2 ** i
            blk = _Holder()
            blk.block = _up_block(_CHANNELS[r], _CHANNELS[r * 2])
            dec.append(blk)
            out_ch = _CHANNELS[r * 2]
        self.decoder_group = _seq(*dec)
        self.out_conv = nn.Conv2d(out_ch, 3, 3, 1, 1)

        # multi-scale vector quantisers (femasr_arch.py:277-300)
        quant, before, after = [], [], []
        for k in range(codebook_params.shape[0]):
            n_e, e_dim = int(codebook_emb_num[k]), int(codebook_emb_dim[k])
            q = _Holder()
            q.embedding = nn.Embedding(n_e, e_dim)
            q.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)       # femasr_arch.py:33
            quant.append(q)
            qc = _CHANNELS[int(self.codebook_scale[k])]
            before.append(nn.Conv2d(qc if k == 0 else 2 * qc, e_dim, 1))
            aq = _Holder()                                              # CombineQuantBlock (fema_utils.py:87-99)
            aq.conv = nn.Conv2d(e_dim if k == 0 else int(codebook_emb_dim[k - 1]) + e_dim, qc, 3, 1, 1)
            after.append(aq)
        self.quantize_group = _seq(*quant)
        self.before_quant_group = _seq(*before)
        self.after_quant_group = _seq(*after)
        self._cb = [(int(s_), int(n), int(d)) for s_, n, d in zip(self.codebook_scale, codebook_emb_num, codebook_emb_dim)]

        for p in self.parameters():
            p.requires_grad_(False)
        self._handle = None
        self._handle_device = None
        self._pushed = {}
        self._weights_dirty = True      # set by load_state_dict / .to() / invalidate_weights(): re-scan the state dict
        self._ws = None
        self.max_tile_batch = 16        # tiles per batched test() call inside test_tile
        self.num_streams = 1            # sub-batch streams inside one forward (femasr_set_streams)
        self._streams_set = None
        # 'fp32' (default): every layer fp32; the convs BEHIND the codebook lookup in the Winograd F(4x4,3x3) form, the SiLU of
        #   their GroupNorm prologue on the hardware exp2 / rcp units (image within ~1e-5 of the oracle, VQ indices bit-exact)
        # 'fp32_strict': the same with the IEEE-exact SiLU - bit-identical to the oracle (OracleNet())
        # 'fp32_direct': every conv in the direct form - bit-identical to OracleNet(winograd=False)
        # 'bf16x3': the convs behind the lookup on the bf16 matrix cores with a 3-term hi/lo split (within the 1e-3 bound)
        self.decoder_math = ignore_kwargs.get('decoder_math', 'fp32')       # (an extension key of this build in `network_g`)
        # arithmetic of the 1x1 convs / nn.Linear layers (Swin qkv / proj / fc1 / fc2, before_quant):
        # 'bf16_split' (default): fp32-grade product on the bf16 matrix pipe - operands split exactly into three bf16 terms, six partial
        #   products, fp32 accumulation; ~3x closer to fp64 than the fp32 chain, bit-identical to OracleNet() (csrc/kernels_gemm_bf16.hip)
        # 'fp32': one fp32 fmaf chain per output on the fp32 MFMA - bit-identical to OracleNet(linear_math='fp32')
        self.linear_math = ignore_kwargs.get('linear_math', 'bf16_split')
        # True: each (shape, mode) class is captured once into a hipGraph (torch.cuda.CUDAGraph around femasr_forward, which
        # is capture-safe: no allocation / synchronisation inside) and replayed; inputs are copied into the graph's static
        # buffer and the outputs are copies of its static outputs.  Only pays when the ~330 launches are host-bound (tiny
        # batches); results are bit-identical.
        self.use_graph = False
        self._graphs = {}
        self.debug_wino_limits = None   # tests only: (log2_total, log2_image) for this net's planner (include/femasr_hip_debug.h)

    # ------------------------------------------------------------------ weight change tracking
    # The native handle holds REPACKED COPIES of the weights.  Changes made through the nn.Module API are seen
    # (load_state_dict, .to()/.cuda()/.float(), any in-place op on a parameter, which bumps its version counter);
    # writes through `.data` (p.data.copy_(...)) are invisible to torch's version counter: call invalidate_weights().
    def invalidate_weights(self):
        """Force every weight to be re-pushed to the native handle on the next forward."""
        self._pushed = {}
        self._weights_dirty = True
        self._graphs = {}

    def _load_from_state_dict(self, *args, **kwargs):
        self._weights_dirty = True
        return super()._load_from_state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._weights_dirty = True
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._weights_dirty = True
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------ native handle
    def _native(self, device):
        """Create / refresh the C handle and push any changed weights."""
        if device.type != 'cuda':
            raise _lib.FemasrError('FeMaSRNet (MI355X build) runs on a GPU device only; move the module and '
                                   'inputs to cuda. There is no CPU fallback.')
        lib = _lib.load()
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is None or self._handle_device != dev_index:
            self._release()
            cfg = _lib.Config()
            cfg.in_channel, cfg.gt_resolution, cfg.lq_stage = self.in_channel, self.gt_res, int(self.LQ_stage)
            cfg.scale_factor, cfg.use_quantize, cfg.use_residual = int(self.scale_factor), int(self.use_quantize), int(self.use_residual)
            cfg.n_codebooks = len(self._cb)
            for k, (sc, n_e, e_dim) in enumerate(self._cb):
                cfg.codebook_scale[k], cfg.n_e[k], cfg.e_dim[k] = sc, n_e, e_dim
            cfg.device = dev_index
            h = ctypes.c_void_p()
            _lib.check(lib.femasr_create(ctypes.byref(cfg), ctypes.byref(h)))
            self._handle, self._handle_device, self._pushed = h, dev_index, {}
            self._weights_dirty = True
        # the walk over the (~480-tensor) state dict only runs when something may have changed; otherwise a cheap scan of
        # the parameters' version counters (in-place updates) decides
        if not self._weights_dirty:
            self._weights_dirty = self._param_stamp() != self._version_sum
        if self._weights_dirty:
            dirty = False
            for key, t in self.state_dict().items():
                if key.endswith('relative_position_index') or key.endswith('attn_mask'):
                    continue
                if t.device.type != 'cuda':
                    raise _lib.FemasrError(f'parameter {key} is on {t.device}; call .to("cuda") on the module')
                stamp = (t.data_ptr(), t._version)
                if self._pushed.get(key) == stamp:
                    continue
                if not dirty:
                    torch.cuda.synchronize(device)
                    dirty = True
                src = t.detach().to(torch.float32).contiguous()
                shape = (ctypes.c_int64 * src.dim())(*src.shape)
                _lib.check(lib.femasr_set_weight(self._handle, key.encode(), _lib.ptr(src), shape, src.dim()))
                self._pushed[key] = stamp
            if dirty:
                _lib.check(lib.femasr_finalize_weights(self._handle))
            self._version_sum = self._param_stamp()
            self._weights_dirty = False
        if self._streams_set != (self._handle.value, self.num_streams, self.decoder_math, self.linear_math, self.debug_wino_limits):
            if self.decoder_math not in ('fp32', 'bf16x3', 'fp32_direct', 'fp32_strict'):
                raise ValueError(f"decoder_math must be 'fp32', 'fp32_strict', 'fp32_direct' or 'bf16x3', got {self.decoder_math!r}")
            if self.linear_math not in ('fp32', 'bf16_split'):
                raise ValueError(f"linear_math must be 'bf16_split' or 'fp32', got {self.linear_math!r}")
            _lib.check(lib.femasr_set_linear_math(self._handle, {'fp32': 0, 'bf16_split': 1}[self.linear_math]))
            _lib.check(lib.femasr_set_streams(self._handle, int(self.num_streams)))
            _lib.check(lib.femasr_set_decoder_math(self._handle, {'fp32': 0, 'bf16x3': 1, 'fp32_direct': 2, 'fp32_strict': 3}[self.decoder_math]))
            if self.debug_wino_limits is not None:
                _lib.check(lib.femasr_debug_set_wino_limits(self._handle, int(self.debug_wino_limits[0]), int(self.debug_wino_limits[1])))
            self._streams_set = (self._handle.value, self.num_streams, self.decoder_math, self.linear_math, self.debug_wino_limits)
        return lib, self._handle

    def _param_stamp(self):
        """Cheap change detector over the parameters: in-place updates bump `_version`; a replaced Parameter object or rebound
        storage (`mod.weight = nn.Parameter(t)`, `p.data = t`, `p.set_()`) changes `data_ptr()`.  Writes through `.data` views
        stay invisible to both: FeMaSRNet.invalidate_weights()."""
        acc = 0
        for p_ in self.parameters():
            acc = (acc * 1000003 + p_._version * 7919 + p_.data_ptr()) & 0xFFFFFFFFFFFFFFFF
        return acc

    def _release(self):
        if getattr(self, '_handle', None) is not None:
            self._graphs = {}             # captured graphs bake pointers into the handle's repacked weights / plans: drop them with it
            _lib.load().femasr_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self._ws

    def enable_profile(self, on=True):
        lib, h = self._native(next(self.parameters()).device)
        _lib.check(lib.femasr_profile_enable(h, int(on)))
        _lib.check(lib.femasr_profile_reset(h))

    def profile(self):
        """{kernel name: (ms, launches, flops, bytes)} since the last reset (HIP events on the launch stream)."""
        lib, h = self._native(next(self.parameters()).device)
        out = {}
        for s in range(lib.femasr_profile_slots(h)):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            _lib.check(lib.femasr_profile_get(h, s, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)))
            if n.value:
                out[lib.femasr_profile_name(h, s).decode()] = (ms.value, n.value, fl.value, by.value)
        return out

    # ------------------------------------------------------------------ the path
    def _launch(self, lib, h, x, pad_mode, oh, ow, sizes, out=None):
        b, _, hh, ww = x.shape
        nbytes = ctypes.c_size_t()
        _lib.check(lib.femasr_workspace_bytes(h, b, hh, ww, pad_mode, ctypes.byref(nbytes)))
        ws = self._workspace(nbytes.value, x.device)
        if out is None:
            out = torch.empty((b, 3, oh, ow), dtype=torch.float32, device=x.device)
        elif (tuple(out.shape) != (b, 3, oh, ow) or out.dtype != torch.float32 or out.device != x.device or not out.is_contiguous()):
            raise ValueError(f'out= must be a contiguous float32 tensor of shape {(b, 3, oh, ow)} on {x.device}, got '
                             f'{tuple(out.shape)} {out.dtype} on {out.device}')
        idx_all = torch.empty((sum(sizes),), dtype=torch.int64, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(lib.femasr_forward(h, ctypes.c_void_p(stream), _lib.ptr(x), b, hh, ww, pad_mode,
                                      _lib.ptr(out), _lib.ptr(idx_all), _lib.ptr(ws), ws.numel()))
        return out, idx_all, ws

    def _run(self, x, pad_mode, out=None):
        if x.dim() != 4 or x.shape[1] != self.in_channel:
            raise ValueError(f'expected (B,{self.in_channel},H,W), got {tuple(x.shape)}')
        lib, h = self._native(x.device)
        x = x.detach().to(torch.float32).contiguous()
        b, _, hh, ww = x.shape
        oh, ow, nq = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        qh, qw = (ctypes.c_int * _lib.MAX_CODEBOOKS)(), (ctypes.c_int * _lib.MAX_CODEBOOKS)()
        _lib.check(lib.femasr_forward_shapes(h, hh, ww, pad_mode, ctypes.byref(oh), ctypes.byref(ow), ctypes.byref(nq),
                                             ctypes.byref(qh), ctypes.byref(qw)))
        sizes = [b * qh[k] * qw[k] for k in range(nq.value)]
        if self.use_graph:
            key = (b, hh, ww, pad_mode, self.num_streams, self.decoder_math, self.linear_math, x.device.index)
            ent = self._graphs.get(key)
            if ent is None:
                if len(self._graphs) > 8:
                    self._graphs.clear()
                xs = x.clone()
                self._launch(lib, h, xs, pad_mode, oh.value, ow.value, sizes)        # warm-up: lazy one-time setup outside capture
                torch.cuda.synchronize(x.device)
                g = torch.cuda.CUDAGraph()
                self._ws = None                                                      # the graph owns its workspace
                with torch.cuda.graph(g):
                    so, si, sw = self._launch(lib, h, xs, pad_mode, oh.value, ow.value, sizes)
                self._ws = None
                ent = self._graphs[key] = (g, xs, so, si, sw)
            g, xs, so, si, _ = ent
            xs.copy_(x)
            g.replay()
            out, idx_all = (so.clone() if out is None else out.copy_(so)), si.clone()
        else:
            out, idx_all, _ = self._launch(lib, h, x, pad_mode, oh.value, ow.value, sizes, out)
        idx, off = [], 0
        for k in range(nq.value):
            idx.append(idx_all[off:off + sizes[k]].view(b, 1, qh[k], qw[k]))
            off += sizes[k]
        return out, idx

    @torch.no_grad()
    def test_u8(self, img_u8, bgr=False, out=None):
        """The CLI arithmetic of inference_femasr.py:50-67 on the device in ONE native call: uint8 (H,W,3) or (B,H,W,3) image(s) ->
        uint8 (sH,sW,3) / (B,sH,sW,3); decode (`/255.`) is fused into the forward's mirror-pad kernel and tensor2img (clamp, x255,
        round half to even) into its crop kernel (femasr_forward_u8) - the same bits as imgproc.u8_to_input -> test() ->
        imgproc.output_to_u8, without the two fp32 NCHW images in between.  `out`: a contiguous uint8 (B,sH,sW,3) tensor the crop
        kernel stores into (the tiled / multi-GPU callers pass slices of their result or all-gather send buffers)."""
        if img_u8.device.type != 'cuda':
            raise _lib.FemasrError('test_u8: tensor must be on the GPU (no CPU fallback)')
        single = img_u8.dim() == 3
        x = img_u8.unsqueeze(0) if single else img_u8
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
            raise ValueError(f'expected uint8 (H,W,3) or (B,H,W,3), got {img_u8.dtype} {tuple(img_u8.shape)}')
        lib, h = self._native(x.device)
        x = x.contiguous()
        b, hh, ww, _ = x.shape
        oh, ow, nq = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        qh, qw = (ctypes.c_int * _lib.MAX_CODEBOOKS)(), (ctypes.c_int * _lib.MAX_CODEBOOKS)()
        _lib.check(lib.femasr_forward_shapes(h, hh, ww, 1, ctypes.byref(oh), ctypes.byref(ow), ctypes.byref(nq), ctypes.byref(qh), ctypes.byref(qw)))
        nbytes = ctypes.c_size_t()
        _lib.check(lib.femasr_workspace_bytes(h, b, hh, ww, 1, ctypes.byref(nbytes)))
        ws = self._workspace(nbytes.value, x.device)
        if out is None:
            out = torch.empty((b, oh.value, ow.value, 3), dtype=torch.uint8, device=x.device)
        elif (out.dtype != torch.uint8 or tuple(out.shape) != (b, oh.value, ow.value, 3) or not out.is_contiguous() or out.device != x.device):
            raise ValueError(f'test_u8(out=...): expected a contiguous uint8 {(b, oh.value, ow.value, 3)} tensor on {x.device}')
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(lib.femasr_forward_u8(h, ctypes.c_void_p(stream), _lib.ptr(x), b, hh, ww, int(bool(bgr)), 1, _lib.ptr(out), None,
                                         _lib.ptr(ws), ws.numel()))
        return out[0] if single else out

    @torch.no_grad()
    def encode_and_decode(self, input, gt_indices=None, current_iter=None):
        """`gt_indices` (one (B,1,h,w) index map per codebook) only changes the LOSS in the reference - VectorQuantizer.forward
        computes z_q and the returned indices from its own argmin either way (femasr_arch.py:64-66,69-91,95,339-342) - so an
        inference build serves the call and ignores it: image and indices are those of forward(input); the losses are zeros."""
        if gt_indices is not None:
            n_cb = len(self._cb)
            if not isinstance(gt_indices, (list, tuple)) or len(gt_indices) < n_cb:
                raise ValueError(f'gt_indices must hold one index map per codebook ({n_cb}), as FeMaSRNet.forward returns them')
        out, idx = self._run(input, 0)
        zero = out.new_zeros(())
        return out, zero, zero, idx

    @torch.no_grad()
    def forward(self, input, gt_indices=None):
        """(dec, codebook_loss, semantic_loss, [indices]) like femasr_arch.py:470-479 (losses are 0 in inference)."""
        return self.encode_and_decode(input, gt_indices)

    @torch.no_grad()
    def test(self, input, out=None):
        """femasr_arch.py:449-468: mirror-pad to (h//wsz+1)*wsz, run, crop to (h*s, w*s).  `out` (not in the reference): a
        contiguous float32 (B, 3, h*s, w*s) tensor the result is written INTO by the last kernel of the forward - the tiled /
        multi-GPU callers pass slices of their result or all-gather send buffers, so no copy follows."""
        return self._run(input, 1, out)[0]

    @torch.no_grad()
    def test_with_indices(self, input):
        """(output, index map of the first codebook) through test()'s pad / crop geometry."""
        out, idx = self._run(input, 1)
        return out, idx[0]

    @torch.no_grad()
    def test_with_all_indices(self, input):
        return self._run(input, 1)

    @torch.no_grad()
    def decode_indices(self, indices):
        assert len(indices.shape) == 4, f'shape of indices must be (b, 1, h, w), but got {indices.shape}'
        dev = next(self.parameters()).device
        lib, h = self._native(dev)
        idx = indices.to(device=dev, dtype=torch.int64).contiguous()
        b, _, qh, qw = idx.shape
        nbytes = ctypes.c_size_t()
        _lib.check(lib.femasr_decode_workspace_bytes(h, b, qh, qw, ctypes.byref(nbytes)))
        ws = self._workspace(nbytes.value, dev)
        up = 2 ** self.max_depth
        out = torch.empty((b, 3, qh * up, qw * up), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.femasr_decode_indices(h, ctypes.c_void_p(stream), _lib.ptr(idx), b, qh, qw, _lib.ptr(out),
                                             _lib.ptr(ws), ws.numel()))
        return out

    # ------------------------------------------------------------------ tiled inference
    @torch.no_grad()
    def test_tile(self, input, tile_size=240, tile_pad=16, rank=0, world_size=1, gather=None, paste=True):
        """Reference semantics of femasr_arch.py:387-447 (overlap-discard paste onto a zero canvas), but
        tiles of one shape class run as batched `test()` calls, and with world_size > 1 each rank
        computes a contiguous share of every class; `gather(results, classes, batch, channel, scale) -> list per rank`
        supplies the collective (femasr_amd.distributed.TileExchange: ONE RCCL all-gather on persistent buffers the tiles were
        written into).  paste=False (ranks other than the one that needs the image): take part in the collective, skip the
        canvas (returns None)."""
        batch, channel, height, width = input.shape
        s = self.scale_factor
        tiles = tiling.enumerate_tiles(height, width, tile_size, tile_pad)
        classes = tiling.shape_classes(tiles)
        owned_all = tiling.assign(classes, world_size, s)          # (once per call: every rank's share - the paste below indexes it)
        mine = owned_all[rank]
        # `time_split = True`: record where the call's time goes (events on the current stream; read back in `last_split_ms`
        # = {'compute', 'gather', 'paste', 'tiles_owned'} after a synchronize) - bench.py --workload tile2048 reports it per rank
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if (getattr(self, 'time_split', False) and input.is_cuda) else None
        if ev:
            ev[0].record()
        # One result buffer per shape class, written in place by the batched test() calls (`out=`).  With a gather object that owns
        # persistent send buffers (femasr_amd.distributed.TileExchange.send_views) the buffers ARE this rank's all-gather send slab.
        alloc = getattr(gather, 'send_views', None) if world_size > 1 else None
        if alloc is not None:
            results = alloc(classes, batch, channel, s, torch.float32, input.device)
        else:
            results = {hw: torch.empty((len(tl) * batch, channel, hw[0] * s, hw[1] * s), dtype=torch.float32, device=input.device)
                       for hw, tl in mine.items()}
        per_call = max(1, self.max_tile_batch // batch)
        in_place = self._test_takes_out()
        for hw, tl in mine.items():
            for i in range(0, len(tl), per_call):
                chunk = tl[i:i + per_call]
                dst = results[hw][i * batch:(i + len(chunk)) * batch]
                crops = self._extract_tiles(input, chunk, hw)
                y = self.test(crops, out=dst) if in_place else self.test(crops)
                if y.data_ptr() != dst.data_ptr():
                    dst.copy_(y)
        if ev:
            ev[1].record()
        if world_size > 1:
            if gather is None:
                raise ValueError('world_size > 1 needs a gather callable')
            per_rank = gather(results, classes, batch, channel, s)
        else:
            per_rank = [results]
        if ev:
            ev[2].record()
        if not paste:
            if ev:
                ev[3].record()
                self._split_events = (ev, sum(len(tl) for tl in mine.values()))
            return None
        output = input.new_zeros((batch, channel, height * s, width * s))
        for r, res in enumerate(per_rank):
            for hw, tl in owned_all[r].items():
                if tl:
                    self._paste_tiles(output, res[hw], tl, batch, s)
        if ev:
            ev[3].record()
            self._split_events = (ev, sum(len(tl) for tl in mine.values()))
        return output

    @torch.no_grad()
    def test_tile_u8(self, img_u8, tile_size=240, tile_pad=16, rank=0, world_size=1, gather=None, paste=True, bgr=False):
        """`test_tile` on uint8 images end to end (round 6; the CLI's tiled branch, inference_femasr.py:58-67 with femasr_arch.py:387-447):
        uint8 (H,W,3) / (B,H,W,3) in -> uint8 (sH,sW,3) / (B,sH,sW,3) out.  Tiles are cropped as uint8, every batched call is ONE
        `test_u8` (decode fused into the mirror-pad kernel, tensor2img into the crop kernel, which stores straight into the result /
        all-gather send buffer), the all-gather and the paste move one byte per value (a quarter of the fp32 path's xGMI and canvas
        traffic; no 805-MB fp32 canvas at 8192^2).  Bit-identical to output_to_u8(test_tile(u8_to_input(img))): tensor2img is
        element-wise and every output pixel comes from exactly one tile."""
        single = img_u8.dim() == 3
        x = img_u8.unsqueeze(0) if single else img_u8
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
            raise ValueError(f'expected uint8 (H,W,3) or (B,H,W,3), got {img_u8.dtype} {tuple(img_u8.shape)}')
        x = x.contiguous()
        batch, height, width, channel = x.shape
        s = self.scale_factor
        tiles = tiling.enumerate_tiles(height, width, tile_size, tile_pad)
        classes = tiling.shape_classes(tiles)
        owned_all = tiling.assign(classes, world_size, s)          # (once: every rank's share, for the paste below)
        mine = owned_all[rank]
        alloc = getattr(gather, 'send_views', None) if world_size > 1 else None
        if alloc is not None:
            results = alloc(classes, batch, channel, s, torch.uint8, x.device, layout='nhwc')
        else:
            results = {hw: torch.empty((len(tl) * batch, hw[0] * s, hw[1] * s, channel), dtype=torch.uint8, device=x.device)
                       for hw, tl in mine.items()}
        per_call = max(1, self.max_tile_batch // batch)
        in_place = 'out' in self._signature_of(self.test_u8)
        for hw, tl in mine.items():
            for i in range(0, len(tl), per_call):
                chunk = tl[i:i + per_call]
                dst = results[hw][i * batch:(i + len(chunk)) * batch]
                crops = self._extract_tiles_u8(x, chunk, hw)
                y = self.test_u8(crops, bgr=bgr, out=dst) if in_place else self.test_u8(crops, bgr=bgr)
                if y.data_ptr() != dst.data_ptr():
                    dst.copy_(y)
        if world_size > 1:
            if gather is None:
                raise ValueError('world_size > 1 needs a gather callable')
            per_rank = gather(results, classes, batch, channel, s, layout='nhwc')
        else:
            per_rank = [results]
        if not paste:
            return None
        output = x.new_zeros((batch, height * s, width * s, channel))
        for r, res in enumerate(per_rank):
            for hw, tl in owned_all[r].items():
                if tl:
                    self._paste_tiles_u8(output, res[hw], tl, batch, s)
        return output[0] if single else output

    @staticmethod
    def _signature_of(fn):
        import inspect
        try:
            return inspect.signature(fn).parameters
        except (TypeError, ValueError):
            return {}

    @staticmethod
    def _extract_tiles_u8(x, chunk, hw):
        """uint8 crops `img[:, y0p:y1p, x0p:x1p, :]` of one shape class as one (n*B, th, tw, 3) batch (tile-major): one native launch
        (host tensors: slicing, for the CPU-side tests of the partition / gather logic)."""
        if not x.is_cuda:
            return torch.cat([x[:, t.y0p:t.y1p, t.x0p:t.x1p, :] for t in chunk], 0).contiguous()
        lib = _lib.load()
        b, h, w, _ = x.shape
        yx = torch.tensor([v for t in chunk for v in (t.y0p, t.x0p)], dtype=torch.int32).to(x.device, non_blocking=True)
        out = torch.empty((len(chunk) * b, hw[0], hw[1], 3), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.femasr_extract_tiles_u8(torch.cuda.current_stream().cuda_stream, x.data_ptr(), b, h, w, yx.data_ptr(),
                                                   len(chunk), hw[0], hw[1], out.data_ptr()))
        return out

    @staticmethod
    def _paste_tiles_u8(output, block, tl, batch, s):
        if not output.is_cuda:
            for k, t in enumerate(tl):
                ys, ye, xs, xe = t.out_src(s)
                dy0, dy1, dx0, dx1 = t.out_dst(s)
                output[:, dy0:dy1, dx0:dx1, :] = block[k * batch:(k + 1) * batch, ys:ye, xs:xe, :]
            return
        lib = _lib.load()
        block = block.contiguous()
        rects, hmax = [], 0
        for t in tl:
            ys, ye, xs, xe = t.out_src(s)
            dy0, _, dx0, _ = t.out_dst(s)
            rects += [ys, xs, dy0, dx0, ye - ys, xe - xs]
            hmax = max(hmax, ye - ys)
        rd = torch.tensor(rects, dtype=torch.int32).to(output.device, non_blocking=True)
        with torch.cuda.device(output.device):
            _lib.check(lib.femasr_paste_tiles_u8(torch.cuda.current_stream().cuda_stream, block.data_ptr(), batch, len(tl), block.shape[1],
                                                 block.shape[2], rd.data_ptr(), hmax, output.shape[1], output.shape[2], output.data_ptr()))

    def _test_takes_out(self):
        """False when `test` was replaced by a stand-in without the `out=` parameter (CPU-side tests of the host logic)."""
        import inspect
        try:
            return 'out' in inspect.signature(self.test).parameters
        except (TypeError, ValueError):
            return False

    @property
    def last_split_ms(self):
        """{'compute', 'gather', 'paste'} milliseconds of the last `test_tile` call made with `time_split = True` (synchronises)."""
        rec = getattr(self, '_split_events', None)
        if rec is None:
            return None
        ev, owned = rec
        ev[3].synchronize()
        return {'compute': round(ev[0].elapsed_time(ev[1]), 3), 'gather': round(ev[1].elapsed_time(ev[2]), 3),
                'paste': round(ev[2].elapsed_time(ev[3]), 3), 'tiles_owned': owned}

    @staticmethod
    def _extract_tiles(input, chunk, hw):
        """torch.cat of the reference's crops `input[:, :, y0p:y1p, x0p:x1p]` (femasr_arch.py:412-426) for the tiles of one
        shape class: one femasr_extract_tiles launch on the GPU (host tensors: plain slicing, used by the CPU-side tests of
        the partition / gather logic only)."""
        if not input.is_cuda:
            return torch.cat([input[:, :, t.y0p:t.y1p, t.x0p:t.x1p] for t in chunk], 0)
        lib = _lib.load()
        b, c, h, w = input.shape
        x = input.contiguous().float()
        yx = torch.tensor([v for t in chunk for v in (t.y0p, t.x0p)], dtype=torch.int32).to(x.device, non_blocking=True)
        out = torch.empty((len(chunk) * b, c, hw[0], hw[1]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.femasr_extract_tiles(torch.cuda.current_stream().cuda_stream, x.data_ptr(), b, c, h, w, yx.data_ptr(),
                                                len(chunk), hw[0], hw[1], out.data_ptr()))
        return out

    @staticmethod
    def _paste_tiles(output, block, tl, batch, s):
        """`output[:, :, dst] = output_tile[:, :, src]` (femasr_arch.py:443-446) for all tiles of one shape class and rank."""
        if not output.is_cuda:
            for k, t in enumerate(tl):
                ys, ye, xs, xe = t.out_src(s)
                dy0, dy1, dx0, dx1 = t.out_dst(s)
                output[:, :, dy0:dy1, dx0:dx1] = block[k * batch:(k + 1) * batch, :, ys:ye, xs:xe]
            return
        lib = _lib.load()
        block = block.contiguous().float()
        rects, hmax = [], 0
        for t in tl:
            ys, ye, xs, xe = t.out_src(s)
            dy0, _, dx0, _ = t.out_dst(s)
            rects += [ys, xs, dy0, dx0, ye - ys, xe - xs]
            hmax = max(hmax, ye - ys)
        rd = torch.tensor(rects, dtype=torch.int32).to(output.device, non_blocking=True)
        with torch.cuda.device(output.device):
            _lib.check(lib.femasr_paste_tiles(torch.cuda.current_stream().cuda_stream, block.data_ptr(), batch, output.shape[1], len(tl),
                                              block.shape[2], block.shape[3], rd.data_ptr(), hmax, output.shape[2], output.shape[3],
                                              output.data_ptr()))
