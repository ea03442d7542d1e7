"""Stock-PyTorch CPU restatement of the FeMaSR inference path (functional, from a plain state dict).

>>> TEST INFRASTRUCTURE ONLY — like everything under oracle/. <<<
Used by bench.py's `cpu_baseline` leg (the "what does this path cost on the host cores with the
reference's own arithmetic library, ATen/oneDNN" number) and by tests/ as a SECOND, independent
checker next to the C oracle (different summation orders, same maths: they must agree to fp32
round-off).  Nothing under femasr_amd/ imports it.

It is NOT the reference's code: /root/reference does not travel to the GPU box.  It is written
against the same state-dict keys from the reference's published semantics:
  pad / crop                    basicsr/archs/femasr_arch.py:449-468
  encoder / decoder / quantise  femasr_arch.py:135-211, 311-374, 35-38, 50-100
  ResBlock / GN(32,1e-6) / SiLU basicsr/archs/fema_utils.py:5-29, 65-84, 87-99
  Swin (RSTB, W-MSA, Mlp)       basicsr/archs/network_swinir.py:14-30, 65-145, 164-279, 419-482
NCHW activations and torch.nn.functional ops throughout, i.e. the kernels a stock PyTorch CPU run of
the reference would spend its time in (mkldnn convolution, addmm, bmm, native group/layer norm).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

_WS = 8          # Swin window size (femasr_arch.py:117)
_HEADS = 8


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


class TorchRefNet:
    """`sd` = {key: ndarray | tensor} with the reference key names; fp32 CPU tensors."""

    def __init__(self, sd, codebook_params=((32, 1024, 512),), gt_resolution=256, LQ_stage=False,
                 scale_factor=4, use_quantize=True, use_residual=True):
        self.sd = {k: _t(v).to(torch.float32) for k, v in sd.items() if not k.endswith(('relative_position_index', 'attn_mask'))}
        self.cb_scales = [int(c[0]) for c in codebook_params]
        self.LQ_stage = bool(LQ_stage)
        self.scale_factor = int(scale_factor) if LQ_stage else 1
        self.gt_res = int(gt_resolution)
        self.use_quantize = use_quantize
        self.use_residual = use_residual
        self.max_depth = int(math.log2(self.gt_res // self.cb_scales[0]))
        self.encode_depth = int(math.log2(self.gt_res // self.scale_factor // self.cb_scales[0]))
        ys, xs = np.divmod(np.arange(_WS * _WS), _WS)
        rel = (ys[:, None] - ys[None, :] + _WS - 1) * (2 * _WS - 1) + (xs[:, None] - xs[None, :] + _WS - 1)
        self._rel_index = torch.from_numpy(rel.reshape(-1).astype(np.int64))
        self._masks = {}
        self.keep_vq_dist = False                    # True: encode_and_decode keeps each lookup's (tokens, n_e) distance matrix in .vq_dist
        self.vq_dist = []

    # ------------------------------------------------------------------ pieces
    def _conv(self, x, p, stride=1, pad=1):
        return F.conv2d(x, self.sd[p + '.weight'], self.sd[p + '.bias'], stride=stride, padding=pad)

    def _gn_silu(self, x, p):
        return F.silu(F.group_norm(x, 32, self.sd[p + '.norm.weight'], self.sd[p + '.norm.bias'], eps=1e-6))

    def _resblock(self, x, p):
        t = self._conv(self._gn_silu(x, p + '.conv.0'), p + '.conv.2')
        t = self._conv(self._gn_silu(t, p + '.conv.3'), p + '.conv.5')
        return x + t

    def _shift_mask(self, h, w, shift):
        key = (h, w, shift)
        if key not in self._masks:
            lab = torch.zeros((h, w))
            cnt = 0
            for a in (slice(0, -_WS), slice(-_WS, -shift), slice(-shift, None)):
                for b in (slice(0, -_WS), slice(-_WS, -shift), slice(-shift, None)):
                    lab[a, b] = cnt
                    cnt += 1
            win = lab.reshape(h // _WS, _WS, w // _WS, _WS).permute(0, 2, 1, 3).reshape(-1, _WS * _WS)
            diff = win[:, None, :] - win[:, :, None]
            self._masks[key] = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))
        return self._masks[key]

    def _swin_block(self, x, h, w, p, shift):
        b, n, c = x.shape
        hd = c // _HEADS
        t = F.layer_norm(x, (c,), self.sd[p + '.norm1.weight'], self.sd[p + '.norm1.bias'], eps=1e-5).reshape(b, h, w, c)
        if shift:
            t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
        win = t.reshape(b, h // _WS, _WS, w // _WS, _WS, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, _WS * _WS, c)
        qkv = F.linear(win, self.sd[p + '.attn.qkv.weight'], self.sd[p + '.attn.qkv.bias'])
        qkv = qkv.reshape(-1, _WS * _WS, 3, _HEADS, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
        att = q @ k.transpose(-2, -1)
        bias = self.sd[p + '.attn.relative_position_bias_table'][self._rel_index].reshape(_WS * _WS, _WS * _WS, _HEADS)
        att = att + bias.permute(2, 0, 1).unsqueeze(0)
        if shift:
            m = self._shift_mask(h, w, shift)
            nw = m.shape[0]
            att = (att.reshape(-1, nw, _HEADS, _WS * _WS, _WS * _WS) + m[None, :, None]).reshape(-1, _HEADS, _WS * _WS, _WS * _WS)
        att = torch.softmax(att, dim=-1)
        o = (att @ v).transpose(1, 2).reshape(-1, _WS * _WS, c)
        o = F.linear(o, self.sd[p + '.attn.proj.weight'], self.sd[p + '.attn.proj.bias'])
        o = o.reshape(b, h // _WS, w // _WS, _WS, _WS, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, c)
        if shift:
            o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
        x = x + o.reshape(b, n, c)
        t = F.layer_norm(x, (c,), self.sd[p + '.norm2.weight'], self.sd[p + '.norm2.bias'], eps=1e-5)
        t = F.gelu(F.linear(t, self.sd[p + '.mlp.fc1.weight'], self.sd[p + '.mlp.fc1.bias']))
        return x + F.linear(t, self.sd[p + '.mlp.fc2.weight'], self.sd[p + '.mlp.fc2.bias'])

    def _swin_layers(self, x, p):
        b, c, h, w = x.shape
        for r in range(4):
            rp = f'{p}.swin_blks.{r}'
            y = x.flatten(2).transpose(1, 2)
            for k in range(6):
                y = self._swin_block(y, h, w, f'{rp}.residual_group.blocks.{k}', 0 if k % 2 == 0 else _WS // 2)
            x = x + self._conv(y.transpose(1, 2).reshape(b, c, h, w), rp + '.conv')
        return x

    def _up_block(self, x, p):
        x = self._conv(F.interpolate(x, scale_factor=2, mode='nearest'), p + '.1')
        return self._resblock(self._resblock(x, p + '.2'), p + '.3')

    def _encoder(self, x):
        p = 'multiscale_encoder'
        x = self._conv(x, p + '.in_conv', 1, 1)
        outs, bi = [], 0
        for _ in range(self.encode_depth):
            x = self._conv(x, f'{p}.blocks.{bi}.0', 2, 1)
            x = self._resblock(self._resblock(x, f'{p}.blocks.{bi}.1'), f'{p}.blocks.{bi}.2')
            outs.append(x)
            bi += 1
        if self.LQ_stage:
            x = self._swin_layers(x, f'{p}.blocks.{bi}')
            outs.append(x)
            bi += 1
            for _ in range(2):
                x = self._up_block(x, f'{p}.blocks.{bi}')
                outs.append(x)
                bi += 1
        return outs

    def _quantize(self, z, qi):
        cb = self.sd[f'quantize_group.{qi}.embedding.weight']
        b, d, h, w = z.shape
        zf = z.permute(0, 2, 3, 1).reshape(-1, d)
        dist = (zf * zf).sum(1, keepdim=True) + (cb * cb).sum(1) - 2.0 * (zf @ cb.t())
        idx = torch.argmin(dist, dim=1)
        if self.keep_vq_dist:                        # bench.py: the reference arithmetic's own distances, to classify index differences
            self.vq_dist.append(dist)
        zq = cb[idx]
        zq = zf + (zq - zf)
        return zq.reshape(b, h, w, d).permute(0, 3, 1, 2).contiguous(), idx.reshape(b, 1, h, w)

    def encode_and_decode(self, x):
        self.vq_dist = []
        feats = self._encoder(x)
        feats = feats[-3:] if self.LQ_stage else feats[::-1]
        fuse_skip = self.LQ_stage and self.use_residual
        indices = []
        prev_q = None
        prev_dec = None
        qi = 0
        x = feats[0]
        for i in range(self.max_depth):
            cur_res = self.gt_res // 2 ** self.max_depth * 2 ** i
            if cur_res in self.cb_scales:            # quantise at this scale (femasr_arch.py:332-359)
                before = feats[i] if prev_dec is None else torch.cat((feats[i], prev_dec), dim=1)
                z = self._conv(before, f'before_quant_group.{qi}', 1, 0)
                if self.use_quantize:
                    zq, idx = self._quantize(z, qi)
                    indices.append(idx)
                else:
                    zq = z
                ap = f'after_quant_group.{qi}.conv'
                if qi == 0:
                    x = self._conv(zq, ap)
                else:                                # CombineQuantBlock with the previous scale (fema_utils.py:92-99)
                    up = F.interpolate(prev_q, zq.shape[2:], mode='nearest')
                    x = self._conv(torch.cat((zq, up), dim=1), ap)
                prev_q = zq
                qi += 1
            elif fuse_skip:                          # femasr_arch.py:361-362
                x = x + feats[i]
            x = self._up_block(x, f'decoder_group.{i}.block')
            prev_dec = x
        return self._conv(x, 'out_conv'), indices

    # ------------------------------------------------------------------ public surface (NCHW)
    @torch.no_grad()
    def forward(self, x):
        out, idx = self.encode_and_decode(_t(x).to(torch.float32))
        return out, idx[0]

    @torch.no_grad()
    def test(self, x, return_indices=False):
        x = _t(x).to(torch.float32)
        _, _, h, w = x.shape
        wsz = 8 // self.scale_factor * 8
        ph, pw = (h // wsz + 1) * wsz - h, (w // wsz + 1) * wsz - w
        x = torch.cat((x, torch.flip(x, [2])), 2)[:, :, :h + ph]
        x = torch.cat((x, torch.flip(x, [3])), 3)[:, :, :, :w + pw]
        out, idx = self.encode_and_decode(x)
        out = out[..., :h * self.scale_factor, :w * self.scale_factor]
        return (out, idx[0]) if return_indices else out
