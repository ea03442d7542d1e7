"""Cases for tools/ubench/mfma_bf16_probe.hip: N x {a[32] bf16, b[32] bf16, c f32} (132-byte records), families tagged in the
returned table.  Every a, b is a bf16 value given as its 16 bits; slot s of a case is k-slot 8 (lane half) + element.

    python tools/ubench/gen_bf16_cases.py          # -> tools/ubench/bf16_cases.bin (not committed; regenerated from the seed)
"""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def bf16_bits(x):
    """float32 array (values exactly representable in bf16) -> uint16 bits"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    assert np.all((u & 0xffff) == 0), 'value not representable in bf16'
    return (u >> 16).astype(np.uint16)


def rne_bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def make_cases(seed=5):
    rng = np.random.default_rng(seed)
    A, B, C, tag = [], [], [], []

    def add(a, b, c, t):
        a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
        aa = np.zeros(32, np.float32); bb = np.zeros(32, np.float32)
        aa[:a.size] = a; bb[:b.size] = b
        A.append(aa); B.append(bb); C.append(np.float32(c)); tag.append(t)

    # F1: random normal, K = 32 slots filled (the 32x32x16 form uses the first 16)
    for i in range(20000):
        add(rne_bf16(rng.standard_normal(32)), rne_bf16(rng.standard_normal(32)), rng.standard_normal() * 4, 'rand')
    for i in range(10000):
        add(rne_bf16(rng.standard_normal(32)), rne_bf16(rng.standard_normal(32)), 0.0, 'rand_c0')
    # F1c: the use case - c large against the products (split terms): products 2^-4 .. 2^-20 of c
    for i in range(10000):
        sc = 2.0 ** -rng.integers(4, 20)
        add(rne_bf16(rng.standard_normal(32)), rne_bf16(rng.standard_normal(32) * sc), rng.standard_normal() * 16, 'rand_small')
    # F2: exponent spread
    for i in range(20000):
        ea = rng.integers(-12, 13, 32); eb = rng.integers(-12, 13, 32)
        ma = 1 + rng.integers(0, 128, 32) / 128.0; mb = 1 + rng.integers(0, 128, 32) / 128.0
        sa = rng.choice([-1.0, 1.0], 32); sb = rng.choice([-1.0, 1.0], 32)
        c = rng.choice([-1.0, 1.0]) * (1 + rng.integers(0, 1 << 23) / float(1 << 23)) * 2.0 ** rng.integers(-24, 25)
        add(sa * ma * 2.0 ** ea, sb * mb * 2.0 ** eb, c, 'spread')
    # F3: cancellation: +2^t at slot p, -2^t at slot q, 1 at slot r (16-slot form; p, q, r < 16), c = 0
    T = [4, 8, 12, 16, 20, 22, 23, 24, 25, 26, 27, 28, 30, 36, 48]
    for t in T:
        for p in range(16):
            for q in range(16):
                if p == q:
                    continue
                for r in rng.choice([x for x in range(16) if x not in (p, q)], 3, replace=False):
                    a = np.zeros(16); b = np.ones(16)
                    a[p] = 2.0 ** t; a[q] = -2.0 ** t; a[r] = 1.0
                    add(a, b, 0.0, f'cancel3_t{t}')
    # F3b: c = 2^t, product -2^t at p, 1 at q
    for t in T:
        for p in range(16):
            for q in range(16):
                if p == q:
                    continue
                a = np.zeros(16); b = np.ones(16)
                a[p] = -2.0 ** t; a[q] = 1.0
                add(a, b, 2.0 ** t, f'cancelc_t{t}')
    # F3c: c = 1 (small), +2^t at p, -2^t at q
    for t in T:
        for p in range(16):
            for q in range(16):
                if p == q:
                    continue
                a = np.zeros(16); b = np.ones(16)
                a[p] = 2.0 ** t; a[q] = -2.0 ** t
                add(a, b, 1.0, f'cancelsmallc_t{t}')
    # F4: rounding of one small product into c
    for c in [1.0, 1.0 + 2.0 ** -23, 1.5, 2.0 - 2.0 ** -23, -1.0, -(1.0 + 2.0 ** -23)]:
        for p in range(16):
            for v in [2.0 ** -24, -2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -25, -3 * 2.0 ** -25, (1 + 2.0 ** -7) * 2.0 ** -24, (1 - 2.0 ** -8) * 2.0 ** -24,
                      2.0 ** -26, 2.0 ** -30, 2.0 ** -40, 5 * 2.0 ** -26, 7 * 2.0 ** -26]:
                a = np.zeros(16); b = np.ones(16); a[p] = v
                add(a, b, c, 'round1')
    # F4b: two small products (tie + sticky; two quarter-ulps making a tie; ...)
    for c in [1.0, 1.0 + 2.0 ** -23, -1.0]:
        for p in range(16):
            for q in range(16):
                if p == q:
                    continue
                for (v, w) in [(2.0 ** -24, 2.0 ** -40), (2.0 ** -24, -2.0 ** -40), (2.0 ** -25, 2.0 ** -25), (2.0 ** -24, 2.0 ** -48), (2.0 ** -24, 2.0 ** -60),
                               (2.0 ** -25, 2.0 ** -26), (3 * 2.0 ** -26, 2.0 ** -26), (2.0 ** -24, 2.0 ** -30)]:
                    a = np.zeros(16); b = np.ones(16); a[p] = v; a[q] = w
                    add(a, b, c, 'round2')
    # F5: n equal small products, c = 1
    for e in [24, 25, 26, 27, 28]:
        for n in range(1, 17):
            for rep in range(4):
                sl = rng.choice(16, n, replace=False)
                a = np.zeros(16); b = np.ones(16); a[sl] = 2.0 ** -e
                add(a, b, 1.0, f'many_e{e}')
    # F6: wide-spread products (how many bits survive?)
    for i in range(5000):
        e = rng.integers(-30, 1, 16)
        a = np.zeros(16); b = np.ones(16); a[:] = 2.0 ** e
        a[rng.integers(0, 16)] = 1.0
        add(a, b, 0.0, 'bits_c0')
    for i in range(5000):
        e = rng.integers(-30, 1, 16)
        a = 2.0 ** e * rng.choice([-1.0, 1.0], 16); b = np.ones(16)
        add(a, b, rng.choice([-1.0, 1.0]) * 2.0 ** rng.integers(-4, 5), 'bits_c')
    if len(A) % 2:
        add(np.zeros(16), np.zeros(16), 0.0, 'pad')
    return np.stack(A), np.stack(B), np.array(C, np.float32), np.array(tag)


def write(path=os.path.join(HERE, 'bf16_cases.bin')):
    A, B, C, tag = make_cases()
    rec = np.zeros(len(C), dtype=np.dtype([('a', '<u2', 32), ('b', '<u2', 32), ('c', '<f4')]))
    rec['a'] = bf16_bits(A); rec['b'] = bf16_bits(B); rec['c'] = C
    assert rec.dtype.itemsize == 132
    rec.tofile(path)
    return len(C)


if __name__ == '__main__':
    print(write(), 'cases')
