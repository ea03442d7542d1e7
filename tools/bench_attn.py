"""Micro-benchmark of the window-attention kernel through the C ABI: python tools/bench_attn.py [B] [H] [W] [shift]."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from femasr_amd import _lib  # noqa: E402

b, h, w, shift = (int(a) for a in (sys.argv[1:5] + ['16', '72', '72', '4'][len(sys.argv) - 1:]))
if os.environ.get('FEMASR_SO'):          # A/B against a debug build
    _lib.SO_PATH = os.environ['FEMASR_SO']
lib = _lib.load()
qkv = torch.randn(b, h * w, 768, device='cuda')
tab = torch.randn(225, 8, device='cuda') * 0.1
out = torch.empty(b, h * w, 256, device='cuda')
for _ in range(3):
    _lib.check(lib.femasr_window_attention(None, _lib.ptr(qkv), b, h, w, 256, 8, shift, _lib.ptr(tab), _lib.ptr(out)))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _lib.check(lib.femasr_window_attention(None, _lib.ptr(qkv), b, h, w, 256, 8, shift, _lib.ptr(tab), _lib.ptr(out)))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f'window_attention B={b} {h}x{w} shift={shift}: {ms * 1e3:.1f} us, {qkv.numel() * 4 / ms / 1e6 + out.numel() * 4 / ms / 1e6:.0f} GB/s')
