"""Micro-benchmark of ONE conv through the C ABI (femasr_conv2d) with device-resident synthetic tensors.
Usage: python tools/bench_conv.py B H W Cin Cout [--up2] [--gn] [--res] [--fp32] [--k1] [--gelu] [--iters N] [--gn-part]
Prints ms per launch and algorithmic TFLOP/s.  With FEMASR_SO=tools/dbg/libfemasr_hip_tt.so (tools/build_debug.sh) it also
prints the per-wave cycle shares and honours FEMASR_BF16_CLS (tile class), FEMASR_DBG / FEMASR_DBG16 (ablation switches)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from femasr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dims', type=int, nargs=5)
    ap.add_argument('--up2', action='store_true')
    ap.add_argument('--gn', action='store_true')
    ap.add_argument('--res', action='store_true')
    ap.add_argument('--fp32', action='store_true')
    ap.add_argument('--wino', action='store_true', help='Winograd F(4x4,3x3) form (fp32)')
    ap.add_argument('--gn-part', action='store_true')
    ap.add_argument('--fast-act', action='store_true', help='Winograd + --gn: SiLU on the hardware exp2 / rcp units (the model default)')
    ap.add_argument('--k1', action='store_true', help='1x1 conv / nn.Linear (fp32 igemm path)')
    ap.add_argument('--gelu', action='store_true')
    ap.add_argument('--iters', type=int, default=10)
    a_ = ap.parse_args()
    b, h, w, cin, cout = a_.dims
    if os.environ.get('FEMASR_SO'):          # debug builds (tools/build_debug.sh)
        _lib.SO_PATH = os.environ['FEMASR_SO']
    lib = _lib.load()
    torch.manual_seed(0)
    dev = 'cuda'
    x = torch.randn(b, h, w, cin, device=dev)
    ks = 1 if a_.k1 else 3
    w_oihw = torch.randn(cout, cin, ks, ks, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    ho, wo = (2 * h, 2 * w) if a_.up2 else (h, w)
    out = torch.empty(b, ho, wo, cout, device=dev)
    wp = torch.empty(int(lib.femasr_packed_weight_floats(cout, cin, ks, ks)), device=dev)
    _lib.check(lib.femasr_repack_oihw(None, _lib.ptr(w_oihw), cout, cin, ks, ks, _lib.ptr(wp)))
    args = _lib.ConvArgs()
    args.in_ = x.data_ptr(); args.B, args.H, args.W, args.Cin = b, h, w, cin
    args.w = wp.data_ptr(); args.bias = bias.data_ptr()
    args.Cout, args.ksz, args.stride, args.pad, args.up2 = cout, ks, 1, (0 if a_.k1 else 1), int(a_.up2)
    args.out = out.data_ptr(); args.Ho, args.Wo = ho, wo
    keep = []
    if a_.gn:
        pa = torch.rand(b, cin, device=dev) + 0.5
        pb = torch.randn(b, cin, device=dev) * 0.1
        args.prologue = _lib.PRO_GN_SILU; args.pro_a = pa.data_ptr(); args.pro_b = pb.data_ptr()
        keep += [pa, pb]
    if a_.gelu:
        args.act = _lib.ACT_GELU
    if a_.res:
        r = torch.randn(b, ho, wo, cout, device=dev)
        args.res1 = r.data_ptr(); keep.append(r)
    if a_.wino:
        if a_.up2:      # the 25-product form of nearest-x2 + conv
            ww = torch.empty(int(lib.femasr_wino_up2_weight_floats(cout, cin)), device=dev)
            _lib.check(lib.femasr_repack_oihw_wino_up2(None, _lib.ptr(w_oihw), cout, cin, _lib.ptr(ww)))
        else:
            ww = torch.empty(int(lib.femasr_wino_weight_floats(cout, cin)), device=dev)
            _lib.check(lib.femasr_repack_oihw_wino(None, _lib.ptr(w_oihw), cout, cin, _lib.ptr(ww)))
        args.w_wino = ww.data_ptr(); keep.append(ww)
        args.fast_act = int(a_.fast_act)
    if not a_.fp32 and not a_.k1 and not a_.wino:
        ws = torch.empty(int(lib.femasr_packed_weight_bf16x3_bytes(cout, cin, 3, 3)), dtype=torch.uint8, device=dev)
        _lib.check(lib.femasr_repack_oihw_bf16x3(None, _lib.ptr(w_oihw), cout, cin, 3, 3, _lib.ptr(ws)))
        args.w_bf16x3 = ws.data_ptr(); keep.append(ws)
    if a_.up2 and a_.fp32 and not a_.wino and cin % 32 == 0:      # phase-filter form of nearest-x2 + conv
        wu = torch.empty(int(lib.femasr_up2_weight_floats(cout, cin)), device=dev)
        _lib.check(lib.femasr_repack_oihw_up2(None, _lib.ptr(w_oihw), cout, cin, _lib.ptr(wu)))
        args.w_up2 = wu.data_ptr(); keep.append(wu)
    if a_.gn_part:
        tiles = ((ho + 15) // 16) * ((wo + 15) // 16) if a_.wino else ((ho + 7) // 8) * ((wo + 15) // 16)
        if a_.up2 and a_.fp32 and not a_.wino:
            tiles = 4 * ((h + 7) // 8) * ((w + 15) // 16)
        part = torch.empty(b, tiles, 32, 2, dtype=torch.float64, device=dev)
        args.gn_part = part.data_ptr(); keep.append(part)
    for _ in range(2):
        _lib.check(lib.femasr_conv2d(None, ctypes.byref(args)))
    torch.cuda.synchronize()
    raw = ctypes.CDLL(_lib.SO_PATH) if os.environ.get('FEMASR_SO') else None
    tt_fn = None
    if raw is not None and a_.wino:
        tt_fn = getattr(raw, 'femasr_debug_wino_up2_time' if a_.up2 else 'femasr_debug_wino_time', None)
    tt = tt_fn is not None
    if tt:
        tt_fn(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a_.iters):
        _lib.check(lib.femasr_conv2d(None, ctypes.byref(args)))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a_.iters
    fl = 2.0 * b * ho * wo * cout * ks * ks * cin
    if tt:
        buf = (ctypes.c_ulonglong * 16)()
        tt_fn(buf, 0)
        tot = float(buf[15]) or 1.0
        if a_.up2:
            names = {0: 'prologue', 1: 'M_phase', 2: 'T_phase', 7: 'barrier', 3: 'epi_write_acc', 4: 'epi_fetch+barrier', 5: 'epi_round', 6: 'epi_barrier2'}
        else:
            names = {0: 'prologue', 1: 'M_phase', 2: 'store_patch', 3: 'transform', 4: 'barrier', 5: 'epi_write_acc', 6: 'epi_fetch+barrier',
                     7: 'epi_round', 8: 'epi_barrier2'}
        print('  per-wave cycle shares: ' + ' '.join('%s=%.3f' % (n, buf[i] / tot) for i, n in sorted(names.items())))
        nsb = b * ((ho + 15) // 16) * ((wo + 15) // 16)
        nwaves = ((nsb + 1) // 2) * (cout // 64) * 8
        print('  cycles per wave: %.0f' % (tot / a_.iters / nwaves))
    print('conv %s dbg=%s cls=%s: %.3f ms  %.1f TFLOP/s (algorithmic)' % (' '.join(sys.argv[1:]), os.environ.get('FEMASR_DBG', '0') + '/' + os.environ.get('FEMASR_DBG16', '0'),
                                                                    os.environ.get('FEMASR_BF16_CLS', '-'), ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()
