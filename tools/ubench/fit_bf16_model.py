"""Fit a summation model to the dump of tools/ubench/mfma_bf16_probe.hip (exact integer arithmetic on the host).

    python tools/ubench/fit_bf16_model.py gpurun_out/bf16_probe_out.bin

Values are held as Python ints at scale 2^-S.  A model maps (products p[0..K), c) -> fp32.
"""
import sys
import os
import math
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_bf16_cases import make_cases  # noqa: E402

S = 400


def f2i(x):
    """float (exactly) -> int at scale 2^-S"""
    x = float(x)
    if x == 0.0:
        return 0
    m, e = math.frexp(x)
    mi = int(m * (1 << 53))
    sh = int(e) - 53 + S
    return mi << sh if sh >= 0 else mi >> (-sh)


def rnd(v, mode='rne', bits=24):
    """int at scale 2^-S -> int at the same scale, rounded to `bits` significant bits (fp32 normal range assumed)"""
    if v == 0:
        return 0
    s = -1 if v < 0 else 1
    a = abs(v)
    n = a.bit_length()
    if n <= bits:
        return v
    sh = n - bits
    q, r = a >> sh, a & ((1 << sh) - 1)
    half = 1 << (sh - 1)
    if mode == 'rne':
        if r > half or (r == half and (q & 1)):
            q += 1
    elif mode == 'rz':
        pass
    elif mode == 'rna':
        if r >= half:
            q += 1
    return s * (q << sh)


def i2f(v):
    return np.float32(float(v) / float(1 << S)) if abs(v) < (1 << 1000) else np.float32(np.inf)


def exact(p, c):
    return rnd(c + sum(p))


def chain(order):
    def f(p, c):
        d = c
        for s in order:
            d = rnd(d + p[s])
        return d
    return f


def grouped(groups, mode='rne'):
    def f(p, c):
        d = c
        for g in groups:
            d = rnd(d + sum(p[s] for s in g), mode)
        return d
    return f


def sum_then_add(p, c):
    return rnd(c + rnd(sum(p)))


def aligned(groups, keep, mode_in='rz', mode_out='rne', c_in_first=True):
    """per group: align every addend (and the running value) to the largest exponent of the group, keep `keep` bits below the
    leading position of that largest addend (truncate), add exactly, round to fp32"""
    def trunc_to(v, top, keep):
        # keep bits down to position top - keep (positions counted from bit_length)
        if v == 0:
            return 0
        sh = top - keep
        if sh <= 0:
            return v
        s = -1 if v < 0 else 1
        a = abs(v)
        if mode_in == 'rz':
            return s * ((a >> sh) << sh)
        # floor (two's complement truncation)
        return (v >> sh) << sh

    def f(p, c):
        d = c
        for g in groups:
            terms = [p[s] for s in g] + [d]
            top = max(abs(t).bit_length() for t in terms)
            acc = sum(trunc_to(t, top, keep) for t in terms)
            d = rnd(acc, mode_out)
        return d
    return f


def main(path):
    A, B, C, tag = make_cases()
    N = len(C)
    out = np.fromfile(path, dtype=np.float32)
    assert out.size == 3 * N, (out.size, N)
    d32, d16, dch = out[:N], out[N:2 * N], out[2 * N:]
    print(N, 'cases')
    P = [[f2i(float(A[n, s]) * float(B[n, s])) for s in range(32)] for n in range(N)]        # bf16 x bf16 is exact in double
    Ci = [f2i(c) for c in C]

    def score(name, model, K, got, sel=None):
        bad = {}
        tot = {}
        idx = range(N) if sel is None else sel
        worst = []
        for n in idx:
            t = tag[n].split('_t')[0]
            tot[t] = tot.get(t, 0) + 1
            want = i2f(model(P[n][:K], Ci[n]))
            if want.view(np.uint32) != got[n].view(np.uint32) and not (want == 0 and got[n] == 0):
                bad[t] = bad.get(t, 0) + 1
                if len(worst) < 3:
                    worst.append((n, tag[n], float(want), float(got[n])))
        nb = sum(bad.values())
        print(f'{name:44s} mismatches {nb:6d} / {sum(tot.values())}   ' + ' '.join(f'{k}:{v}/{tot[k]}' for k, v in sorted(bad.items())))
        return nb, worst

    nat = list(range(16))
    inter = [x for pr in zip(range(8), range(8, 16)) for x in pr]          # 0,8,1,9,...
    for K, got, label in ((16, d32, '32x32x16'), (32, d16, '16x16x32')):
        print('==', label)
        models = {
            'exact sum, one RNE': exact,
            'RNE(c + RNE(sum))': sum_then_add,
            'fma chain, slots ascending': chain(list(range(K))),
            'groups of 2': grouped([list(range(i, i + 2)) for i in range(0, K, 2)]),
            'groups of 4': grouped([list(range(i, i + 4)) for i in range(0, K, 4)]),
            'groups of 8': grouped([list(range(i, i + 8)) for i in range(0, K, 8)]),
            'groups of 16': grouped([list(range(i, i + 16)) for i in range(0, K, 16)]),
            'exact sum, RZ': lambda p, c: rnd(c + sum(p), 'rz'),
        }
        if K == 16:
            models['groups of 4, interleaved halves'] = grouped([[0, 1, 2, 3], [8, 9, 10, 11], [4, 5, 6, 7], [12, 13, 14, 15]])
            models['groups {e, e+8} pairs'] = grouped([[e, e + 8] for e in range(8)])
            models['groups {0-3,8-11},{4-7,12-15}'] = grouped([[0, 1, 2, 3, 8, 9, 10, 11], [4, 5, 6, 7, 12, 13, 14, 15]])
        for keep in (24, 25, 26, 27, 28, 30, 32, 48):
            models[f'aligned all, keep {keep} rz'] = aligned([list(range(K))], keep)
        res = {}
        for name, m in models.items():
            res[name] = score(name, m, K, got)
        best = min(res, key=lambda k: res[k][0])
        print('best:', best, res[best][1])
    # chained hand-over: d = mfma(case n + N/2 products, mfma(case n)) - is the intermediate a plain fp32?
    print('== chain of two 32x32x16 on one accumulator vs exact-per-instruction model')
    N2 = N // 2
    badc = 0
    for n in range(N2):
        mid = exact(P[n][:16], Ci[n])
        want = i2f(exact(P[n + N2][:16], mid))
        if want.view(np.uint32) != dch[n].view(np.uint32) and not (want == 0 and dch[n] == 0):
            badc += 1
    print('exact/exact chained mismatches', badc, '/', N2)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/bf16_probe_out.bin')
