"""The ONE near-tie rule of the parity checks (tests/, tools/parity_report.py, bench.py's vq_index_match).

>>> TEST INFRASTRUCTURE ONLY (like everything under oracle/). <<<

VQ indices are compared with the REFERENCE's own index maps.  The reference's fp32 distances d = |z|^2 + |e|^2 - 2 z.e
(femasr_arch.py:35-38) put the best and the second-best code of a few tokens per million within a few ulp of each other
(|z|^2 ~ 500: one ulp of d is 3e-5 absolute, 6e-8 relative) - closer than the summation-order differences of ANY two correct
fp32 evaluations of the encoder (the reference's own results move with the thread count of its BLAS).  A token may therefore
resolve to another code than the reference's ONLY where the reference itself has that code within NEAR_TIE_ULP ulp of its best
distance; everything else is a mismatch.  Measured (profiles/r05_parity_report.txt): 1 286 016 tokens of the reference's testset/:
a handful of accepted flips, gaps 0-3 ulp.
"""
import numpy as np

NEAR_TIE_ULP = 4.0            # largest accepted gap, in ulp of the reference's best distance
MAX_FLIPS_PER_IMAGE = 4       # accepted flips per test() call of the tiled / testset fixtures
MAX_FLIPS_TESTSET = 24        # ... over the 1 286 016 tokens of the 38-image testset fixture


def gap_ulp(d_best, d_other):
    """Gap of two fp32 distances in ulp of the smaller one."""
    d_best = np.float32(d_best)
    return float((np.float32(d_other) - d_best) / np.spacing(np.abs(d_best)))


def histogram(gaps):
    """{'0': n, '(0,1]': n, ... '(3,4]': n, '>4': n} of accepted / rejected gaps (ulp)."""
    h = {'0': 0, '(0,1]': 0, '(1,2]': 0, '(2,3]': 0, '(3,4]': 0, '>4': 0}
    for g in gaps:
        if g <= 0:
            h['0'] += 1
        elif g > 4:
            h['>4'] += 1
        else:
            h[f'({int(np.ceil(g)) - 1},{int(np.ceil(g))}]'] += 1
    return h
