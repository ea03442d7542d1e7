"""Codebook lookup alone on the GPU: single-pass fp32 MFMA (femasr_vq) vs two-pass exact search (femasr_vq_twopass),
per-stage times and the candidate statistics of pass 1.  python tools/bench_vq.py [--m 82944] [--regime trained|init]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femasr_amd import _lib  # noqa: E402


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=82944)
    ap.add_argument('--d', type=int, default=512)
    ap.add_argument('--ne', type=int, default=1024)
    ap.add_argument('--regime', default='both')
    a = ap.parse_args()
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    for regime in (['trained', 'init', 'gauss'] if a.regime == 'both' else [a.regime]):
        g = torch.Generator(device='cuda').manual_seed(1)
        if regime == 'gauss':        # activations like the encoder's (roughly normal), codes at their scale
            z = torch.randn((a.m, a.d), device='cuda', generator=g)
            cb = torch.randn((a.ne, a.d), device='cuda', generator=g)
        else:
            sc = 1.0 if regime == 'trained' else 1.0 / a.ne
            z = torch.rand((a.m, a.d), device='cuda', generator=g) * 2 - 1
            cb = (torch.rand((a.ne, a.d), device='cuda', generator=g) * 2 - 1) * sc
        cbt = torch.empty(int(lib.femasr_packed_weight_floats(a.ne, a.d, 1, 1)), device='cuda')
        ee = torch.empty(a.ne, device='cuda')
        zz = torch.empty(a.m, device='cuda')
        _lib.check(lib.femasr_repack_oihw(s, _lib.ptr(cb), a.ne, a.d, 1, 1, _lib.ptr(cbt)))
        _lib.check(lib.femasr_row_sqsum(s, _lib.ptr(cb), a.ne, a.d, _lib.ptr(ee)))
        aux = torch.empty(int(lib.femasr_vq_aux_bytes(a.ne, a.d)), dtype=torch.uint8, device='cuda')
        _lib.check(lib.femasr_vq_prepare(s, _lib.ptr(cb), _lib.ptr(ee), a.ne, a.d, _lib.ptr(aux)))
        scratch = torch.empty(int(lib.femasr_vq_scratch_bytes(a.m, a.ne)), dtype=torch.uint8, device='cuda')
        idx1 = torch.empty(a.m, dtype=torch.int64, device='cuda')
        idx2 = torch.empty(a.m, dtype=torch.int64, device='cuda')
        zq1 = torch.empty_like(z)
        zq2 = torch.empty_like(z)
        cand = torch.zeros((a.m, 32), dtype=torch.int16, device='cuda')
        cnt = torch.zeros(a.m, dtype=torch.int16, device='cuda')

        def one():
            _lib.check(lib.femasr_vq(s, _lib.ptr(z), a.m, a.d, _lib.ptr(cb), _lib.ptr(cbt), _lib.ptr(ee), a.ne, _lib.ptr(idx1),
                                     _lib.ptr(zq1), _lib.ptr(scratch)))

        def two():
            _lib.check(lib.femasr_vq_twopass(s, _lib.ptr(z), a.m, a.d, _lib.ptr(cb), _lib.ptr(aux), _lib.ptr(ee), a.ne,
                                             _lib.ptr(idx2), _lib.ptr(zq2), _lib.ptr(scratch)))

        def sq():
            _lib.check(lib.femasr_row_sqsum(s, _lib.ptr(z), a.m, a.d, _lib.ptr(zz)))

        def p1():
            _lib.check(lib.femasr_vq_candidates(s, _lib.ptr(z), a.m, a.d, _lib.ptr(aux), _lib.ptr(ee), a.ne, _lib.ptr(cand), _lib.ptr(cnt)))

        t1, t2, tsq, tp1 = timed(one), timed(two), timed(sq), timed(p1)
        same = bool((idx1 == idx2).all()) and bool((zq1.view(torch.int32) == zq2.view(torch.int32)).all())
        c = cnt.cpu().numpy().view('uint16').astype(np.int64)
        every = c == 0xFFFF
        hist = np.bincount(c[~every], minlength=33)
        print(f'[{regime}] M={a.m} D={a.d} n_e={a.ne}: single-pass {t1:.0f} us | two-pass {t2:.0f} us '
              f'(candidates {tp1:.0f}, exact+write {t2 - tp1:.0f}; single-pass row_sqsum {tsq:.0f}) | identical={same}')
        print(f'    candidates/row mean {c[~every].mean():.2f} max {c[~every].max()} "every code" rows {int(every.sum())}; '
              f'hist 1..8 {hist[1:9].tolist()} >8 {int(hist[9:].sum())}')
        flops = 2.0 * a.m * a.ne * a.d
        print(f'    pass-1 bf16 MFMA rate {flops / tp1 / 1e6:.1f} TFLOP/s; z bytes {a.m * a.d * 4 / 1e6:.0f} MB')


if __name__ == '__main__':
    main()
