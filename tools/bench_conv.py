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


def tt_report(T, nsteps, floor=4608):
    """Cycle stamps of the last launch (tools/build_debug.sh tt): T[block][wave 0 | 7][slot], slots as in kernels_wino.hip."""
    import numpy as np
    T = T.astype(np.float64)
    w0 = T[:, 0]
    cyc = w0[:, 10] - w0[:, 63]
    rt = (w0[:, 11] - w0[:, 62]) * 10.0                      # s_memrealtime: 100 MHz -> ns
    ghz = float(np.median(cyc / np.maximum(rt, 1.0)))
    span_ns = (T[:, :, 11].max() - T[:, :, 62].min()) * 10.0
    us = lambda c: c / ghz / 1e3
    ns = min(nsteps, 40)
    steps = np.diff(np.concatenate([w0[:, 0:1], w0[:, 16:16 + ns]], axis=1), axis=1)
    print('  clock %.3f GHz; %d blocks; kernel span %.1f us; block life mean %.1f us (min %.1f, max %.1f); sum(block life) / (span x 256 CUs) = %.3f'
          % (ghz, len(cyc), span_ns / 1e3, us(cyc.mean()), us(cyc.min()), us(cyc.max()), rt.sum() / (span_ns * 256)))
    ph = [('prologue', w0[:, 0] - w0[:, 63]), ('main loop', w0[:, 1] - w0[:, 0])]
    for r in range(2):
        base = w0[:, 1] if r == 0 else w0[:, 5]
        ph += [('r%d acc->LDS' % r, w0[:, 2 + 4 * r] - base), ('r%d fetch+barrier' % r, w0[:, 3 + 4 * r] - w0[:, 2 + 4 * r]),
               ('r%d transform+store' % r, w0[:, 4 + 4 * r] - w0[:, 3 + 4 * r]), ('r%d barrier2' % r, w0[:, 5 + 4 * r] - w0[:, 4 + 4 * r])]
    ph += [('tail (moments out)', w0[:, 10] - w0[:, 9])]
    for n, v in ph:
        print('    %-22s %8.2f us  %5.1f %%   (min %.2f max %.2f)' % (n, us(v.mean()), 100 * v.mean() / cyc.mean(), us(v.min()), us(v.max())))
    print('    per step: mean %.3f us (%.0f cycles; first %.0f, median step %.0f, last %.0f; MFMA floor %d)'
          % (us(steps.mean()), steps.mean(), steps[:, 0].mean(), np.median(steps.mean(axis=0)), steps[:, -1].mean(), floor))
    if T.shape[2] > 64:        # inside the steps: M phase issued / T phase done / barrier passed
        n2 = min(ns, 24)
        m_end, t_end = w0[:, 64:64 + 2 * n2:2], w0[:, 65:65 + 2 * n2:2]
        start = np.concatenate([w0[:, 0:1], w0[:, 16:16 + n2 - 1]], axis=1)
        bar = w0[:, 16:16 + n2]
        print('    inside a step (cycles, mean over blocks and steps): M phase %.0f   T phase %.0f   barrier wait %.0f   | wave 7: M %.0f T %.0f barrier %.0f'
              % ((m_end - start).mean(), (t_end - m_end).mean(), (bar - t_end).mean(),
                 (T[:, 1, 64:64 + 2 * n2:2] - np.concatenate([T[:, 1, 0:1], T[:, 1, 16:16 + n2 - 1]], axis=1)).mean(),
                 (T[:, 1, 65:65 + 2 * n2:2] - T[:, 1, 64:64 + 2 * n2:2]).mean(), (T[:, 1, 16:16 + n2] - T[:, 1, 65:65 + 2 * n2:2]).mean()))
    d7 = T[:, 1, 10] - T[:, 0, 10]
    print('    wave 7 ends %.2f us after wave 0 on average (|max| %.2f)' % (us(d7.mean()), us(np.abs(d7).max())))
    st = np.sort((w0[:, 62] - w0[:, 62].min()) * 10.0 / 1e3)
    print('    block starts (us from the first): 10%% %.1f  50%% %.1f  90%% %.1f  last %.1f' % (st[len(st) // 10], st[len(st) // 2], st[len(st) * 9 // 10], st[-1]))


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
    ap.add_argument('--bf16m', action='store_true', help='--wino --gn --fast-act + the M phase on the bf16 matrix pipe (fast_act = 2, round 5)')
    ap.add_argument('--k1', action='store_true', help='1x1 conv / nn.Linear (fp32 igemm path)')
    ap.add_argument('--gelu', action='store_true')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--power', action='store_true', help='sample package power / engine clock (amdgpu hwmon) while the timed loop runs: use a few thousand iterations')
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
        args.fast_act = 2 if a_.bf16m else int(a_.fast_act)
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
    raw = ctypes.CDLL(_lib.SO_PATH) if os.environ.get('FEMASR_SO') else None
    tt_fn = None
    c128 = False          # (the 16x16 x 128 block shape of round 4 was removed in round 6)
    if raw is not None and a_.wino:
        tt_fn = getattr(raw, 'femasr_debug_wino_up2_ttbuf' if a_.up2 else 'femasr_debug_wino_ttbuf', None)
    tt = tt_fn is not None
    if tt:
        nsb = b * ((ho + 15) // 16) * ((wo + 15) // 16)
        nblk = nsb * (cout // 128) if c128 else ((nsb + 1) // 2) * (cout // 64)
        TTS = 64 if a_.up2 else 128
        ttbuf = torch.zeros(nblk * 2 * TTS, dtype=torch.int64, device=dev)
        tt_fn.argtypes = [ctypes.c_void_p]
        assert tt_fn(ttbuf.data_ptr()) == 0
    for _ in range(2):
        _lib.check(lib.femasr_conv2d(None, ctypes.byref(args)))
    torch.cuda.synchronize()
    watch = None
    if a_.power:
        import bench
        pr = torch.cuda.get_device_properties(0)
        watch = bench.PowerWatch((pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
        watch.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a_.iters):
        _lib.check(lib.femasr_conv2d(None, ctypes.byref(args)))
    e1.record()
    torch.cuda.synchronize()
    if watch:
        pw = watch.stop()
        print('  power:', {k: v for k, v in (pw or {}).items() if k in ('package_w_mean', 'package_w_max', 'sclk_mhz_median', 'samples')})
    ms = e0.elapsed_time(e1) / a_.iters
    fl = 2.0 * b * ho * wo * cout * ks * ks * cin
    if tt:
        tt_report(ttbuf.cpu().numpy().reshape(nblk, 2, TTS), cin // 16 if c128 else cin // 8, 9216 if c128 else 4608)
    print('conv %s dbg=%s cls=%s: %.3f ms  %.1f TFLOP/s (algorithmic)' % (' '.join(sys.argv[1:]), os.environ.get('FEMASR_DBG', '0') + '/' + os.environ.get('FEMASR_DBG16', '0'),
                                                                    os.environ.get('FEMASR_BF16_CLS', '-'), ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()
