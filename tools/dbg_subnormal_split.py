"""Diagnostic (GPU): where does the split GEMM's own operand split differ from orc_split3 on tiny inputs?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gpu_utils as G
from oracle import oracle as orc

def bfval(p): return (p.astype(np.uint32) << 16).view(np.float32)

rng = np.random.default_rng(8)
rows, cin, cout = 256, 64, 32
w = rng.standard_normal((cout, cin)).astype(np.float32)
b = np.zeros(cout, np.float32)
for lo, hi in ((-100, -90), (-110, -100), (-118, -110), (-126, -118), (-134, -126), (-149, -134)):
    x = (rng.standard_normal((rows, cin)) * np.exp2(rng.integers(lo, hi, (rows, 1)).astype(np.float64))).astype(np.float32)
    y = G.conv2d(x.reshape(1, rows, 1, cin), np.ascontiguousarray(w.T).reshape(1, 1, cin, cout), b, 1, bf16s=True).reshape(rows, cout)
    yo = orc.linear_bf16s(x, w, b)
    p = orc.split3(x)
    sub = [((pp & 0x7f80) == 0) & ((pp & 0x7f) != 0) for pp in p]
    # hypothesis A: subnormal bf16 terms flushed to zero by the kernel's conversion
    pa = [np.where(s, 0, pp).astype(np.uint16) for s, pp in zip(sub, p)]
    xa = (bfval(pa[0]).astype(np.float64) + bfval(pa[1]) + bfval(pa[2])).astype(np.float32)
    ya = orc.linear_bf16s(xa, w, b)
    # hypothesis B: fp32 subnormal remainders flushed before the conversion (x - x1 subnormal -> 0)
    x1 = bfval(p[0]); r = (x - x1).astype(np.float32); r[np.abs(r) < 2.0 ** -126] = 0
    p2 = orc.split3(r)[0]; s = (r - bfval(p2)).astype(np.float32); s[np.abs(s) < 2.0 ** -126] = 0
    p3 = orc.split3(s)[0]
    xb = (x1.astype(np.float64) + bfval(p2) + bfval(p3)).astype(np.float32)
    yb = orc.linear_bf16s(xb, w, b)
    # hypothesis C: fp32 subnormal INPUT x flushed
    xc = x.copy(); xc[np.abs(xc) < 2.0 ** -126] = 0
    yc = orc.linear_bf16s(xc, w, b)
    f = lambda u: int((u.view(np.uint32) != y.view(np.uint32)).sum())
    print(f'|x| ~ 2^[{lo},{hi}): subnormal terms p1/p2/p3 {[int(s.sum()) for s in sub]} of {x.size}; differ: oracle {f(yo)}, A(flush bf16 subnormal terms) {f(ya)}, '
          f'B(flush fp32 subnormal remainders) {f(yb)}, C(flush subnormal x) {f(yc)} of {y.size}; gpu zeros {int((y == 0).sum())} oracle zeros {int((yo == 0).sum())}')
