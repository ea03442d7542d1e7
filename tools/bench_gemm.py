"""Micro-benchmark of the Swin linears through the C ABI (femasr_conv2d, 1x1) in every block configuration of kernels_gemm.hip
(femasr_gemm_force_config): ms per launch and TFLOP/s for the token counts of B = 1, 2, 4, 8, 16 tiles of 128x128 (72x72 tokens each).
Usage: python tools/bench_gemm.py [--iters N]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from femasr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batches', type=int, nargs='*', default=[1, 2, 4, 8, 16])
    a_ = ap.parse_args()
    lib = _lib.load()
    torch.manual_seed(0)
    dev = 'cuda'
    layers = [('qkv', 256, 768, False, False), ('proj', 256, 256, False, True), ('fc1', 256, 1024, True, False), ('fc2', 1024, 256, False, True)]
    names = {-1: 'auto', 0: '128/k32', 1: '128/k16', 2: '64x64'}
    for B in a_.batches:
        M = B * 72 * 72
        for lname, cin, cout, gelu, res in layers:
            x = torch.randn(1, M, 1, cin, device=dev)
            w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
            bias = torch.randn(cout, device=dev)
            out = torch.empty(1, M, 1, cout, device=dev)
            wp = torch.empty(int(lib.femasr_packed_weight_floats(cout, cin, 1, 1)), device=dev)
            _lib.check(lib.femasr_repack_oihw(None, _lib.ptr(w), cout, cin, 1, 1, _lib.ptr(wp)))
            args = _lib.ConvArgs()
            args.in_ = x.data_ptr(); args.B, args.H, args.W, args.Cin = 1, M, 1, cin
            args.w = wp.data_ptr(); args.bias = bias.data_ptr()
            args.Cout, args.ksz, args.stride, args.pad, args.up2 = cout, 1, 1, 0, 0
            args.out = out.data_ptr(); args.Ho, args.Wo = M, 1
            if gelu:
                args.act = _lib.ACT_GELU
            if res:
                r = torch.randn(1, M, 1, cout, device=dev)
                args.res1 = r.data_ptr()
            line = f'B={B:2d} M={M:6d} {lname:4s} {cin:4d}->{cout:4d}:'
            ref = None
            for cfg in (-1, 0, 1, 2):
                lib.femasr_gemm_force_config(cfg)
                for _ in range(3):
                    _lib.check(lib.femasr_conv2d(None, ctypes.byref(args)))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a_.iters):
                    _lib.check(lib.femasr_conv2d(None, ctypes.byref(args)))
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a_.iters
                same = ''
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    same = ' MISMATCH'
                line += f'  {names[cfg]} {ms * 1e3:7.1f} us ({2.0 * M * cin * cout / ms / 1e9:5.1f} TF){same}'
            lib.femasr_gemm_force_config(-1)
            print(line, flush=True)


if __name__ == '__main__':
    main()
