"""Micro-benchmark of the Swin MLP: the fused kernel (femasr_mlp_fused) against the two GEMM launches it replaces (fc1 + GELU, fc2 + residual).
Usage: python tools/bench_mlp.py [rows ...]      (default: 82944 41472 5184 - B = 16 / 8 / 1 of 72x72 tokens)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from femasr_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = 'cuda'
    C, Hd = 256, 1024
    torch.manual_seed(0)
    w1 = torch.randn(Hd, C, device=dev) * 0.05
    w2 = torch.randn(C, Hd, device=dev) * 0.03
    b1, b2 = torch.randn(Hd, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    w1p = torch.empty(int(lib.femasr_packed_weight_floats(Hd, C, 1, 1)), device=dev)
    w2p = torch.empty(int(lib.femasr_packed_weight_floats(C, Hd, 1, 1)), device=dev)
    _lib.check(lib.femasr_repack_oihw(None, _lib.ptr(w1), Hd, C, 1, 1, _lib.ptr(w1p)))
    _lib.check(lib.femasr_repack_oihw(None, _lib.ptr(w2), C, Hd, 1, 1, _lib.ptr(w2p)))
    for rows in [int(a) for a in sys.argv[1:]] or [82944, 41472, 5184]:
        x = torch.randn(rows, C, device=dev)
        res = torch.randn(rows, C, device=dev)
        out = torch.empty(rows, C, device=dev)
        hid = torch.empty(rows, Hd, device=dev)

        def fused():
            _lib.check(lib.femasr_mlp_fused(None, _lib.ptr(x), rows, C, Hd, _lib.ptr(w1p), _lib.ptr(b1), _lib.ptr(w2p), _lib.ptr(b2), _lib.ptr(res), _lib.ptr(out)))

        a1, a2 = _lib.ConvArgs(), _lib.ConvArgs()
        for a, (i, o, ci, co, w, b) in ((a1, (x, hid, C, Hd, w1p, b1)), (a2, (hid, out, Hd, C, w2p, b2))):
            a.in_ = i.data_ptr(); a.B, a.H, a.W, a.Cin = 1, rows, 1, ci
            a.w = w.data_ptr(); a.bias = b.data_ptr(); a.Cout, a.ksz, a.stride, a.pad = co, 1, 1, 0
            a.out = o.data_ptr(); a.Ho, a.Wo = rows, 1
        a1.act = _lib.ACT_GELU
        a2.res1 = res.data_ptr()

        def split():
            _lib.check(lib.femasr_conv2d(None, ctypes.byref(a1)))
            _lib.check(lib.femasr_conv2d(None, ctypes.byref(a2)))

        fl = 2.0 * rows * 2 * C * Hd
        for name, fn in (('fused', fused), ('two launches', split)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f'rows {rows:6d} {name:13s}: {ms:.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s')


if __name__ == '__main__':
    main()
