"""Corner families of v_mfma_f32_32x32x16_bf16 that the round-5 fixture did not cover: bf16 SUBNORMAL operands,
third split terms that underflow, +-0 accumulators / results, fp32-subnormal accumulators and results, Inf / NaN propagation.

    gpurun -- 'python tools/mfma_corner_probe.py gpurun_out/mfma_bf16_corners.npz'

Every case = (16 bf16 products a[s] * b[s], fp32 accumulator c) -> d, run on the hardware through femasr_debug_mfma_bf16 and through
the oracle's restatement (two sequential orc_mfma_dot8 groups).  Prints, per family, how many results differ (bit patterns; NaN
payloads compared as "both NaN") and writes the cases WITH THE HARDWARE's answers: the fixture the restatement is then pinned on
(tests/golden/mfma_bf16_corners.npz, tests/test_oracle_units.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def bf(x):
    """fp32 array -> bf16 bit patterns (truncation of an exactly representable value)."""
    return (np.asarray(x, np.float32).view(np.uint32) >> 16).astype(np.uint16)


def cases(seed=3):
    rng = np.random.default_rng(seed)
    fam, A, B, C = [], [], [], []

    def add(name, a, b, c):
        a, b = np.asarray(a, np.uint16).reshape(-1, 16), np.asarray(b, np.uint16).reshape(-1, 16)
        c = np.asarray(c, np.float32).reshape(-1)
        assert len(a) == len(b) == len(c)
        fam.extend([name] * len(c)); A.append(a); B.append(b); C.append(c)

    def rnd_norm(n, lo=-6, hi=6):           # random normal bf16 values with exponents in [lo, hi)
        m = rng.integers(0, 128, (n, 16)).astype(np.uint16)
        e = (rng.integers(lo, hi, (n, 16)) + 127).astype(np.uint16)
        s = rng.integers(0, 2, (n, 16)).astype(np.uint16)
        return (s << 15) | (e << 7) | m

    n = 400
    # --- bf16 subnormal operands (exponent field 0, mantissa != 0): flushed to zero, or multiplied?
    sub = (rng.integers(0, 2, (n, 16)).astype(np.uint16) << 15) | rng.integers(1, 128, (n, 16)).astype(np.uint16)
    big = rnd_norm(n, 100, 126)             # partners large enough that sub * big is a normal fp32 number
    add('sub_a_times_big', sub, big, np.zeros(n))
    add('sub_b_times_big', big, sub, np.zeros(n))
    mix = np.where(rng.random((n, 16)) < 0.3, sub, rnd_norm(n))
    add('sub_mixed_acc', mix, rnd_norm(n), rng.standard_normal(n))
    one_sub = rnd_norm(n) * 0
    one_sub[:, 0] = sub[:, 0]
    one_big = rnd_norm(n) * 0
    one_big[:, 0] = big[:, 0]
    add('one_sub_product_only', one_sub, one_big, np.zeros(n))
    add('one_sub_product_acc', one_sub, one_big, (rng.standard_normal(n) * 1e-3).astype(np.float32))
    # --- tiny products: results below the fp32 normal range (flush? denormal result?)
    ta, tb = rnd_norm(n, -70, -60), rnd_norm(n, -70, -60)          # products ~2^-130: below 2^-126
    add('tiny_products_zero_acc', ta, tb, np.zeros(n))
    add('tiny_products_tiny_acc', ta, tb, (rng.standard_normal(n) * 2.0 ** -125).astype(np.float32))
    ta2, tb2 = rnd_norm(n, -66, -60), rnd_norm(n, -66, -60)        # products ~2^-126 .. 2^-120: straddling the boundary
    add('boundary_products', ta2, tb2, np.zeros(n))
    # --- fp32 subnormal accumulator
    subacc = (rng.integers(1, 1 << 23, n).astype(np.uint32) | (rng.integers(0, 2, n).astype(np.uint32) << 31)).view(np.float32)
    add('subnormal_acc_zero_products', np.zeros((n, 16), np.uint16), np.zeros((n, 16), np.uint16), subacc)
    add('subnormal_acc_small_products', rnd_norm(n, -66, -62), rnd_norm(n, -66, -62), subacc)
    add('subnormal_acc_normal_products', rnd_norm(n), rnd_norm(n), subacc)
    # --- signed zeros
    z = np.zeros((8, 16), np.uint16)
    nz = np.full((8, 16), 0x8000, np.uint16)
    one = np.full((8, 16), 0x3f80, np.uint16)
    add('zero_products_pos_acc0', z, one, np.zeros(8))
    add('zero_products_neg_acc0', z, one, np.full(8, -0.0, np.float32))
    add('negzero_products_pos_acc0', nz, one, np.zeros(8))
    add('negzero_products_neg_acc0', nz, one, np.full(8, -0.0, np.float32))
    canc_a = np.zeros((8, 16), np.uint16); canc_b = np.zeros((8, 16), np.uint16)
    canc_a[:, 0], canc_b[:, 0], canc_a[:, 1], canc_b[:, 1] = 0x3f80, 0x4000, 0xbf80, 0x4000          # 1*2 + (-1)*2 = 0 inside group 0
    add('exact_cancellation_acc0', canc_a, canc_b, np.zeros(8))
    add('exact_cancellation_negacc0', canc_a, canc_b, np.full(8, -0.0, np.float32))
    canc2_a, canc2_b = canc_a.copy(), canc_b.copy()
    canc2_a[:, 1], canc2_b[:, 1] = 0, 0
    canc2_a[:, 8], canc2_b[:, 8] = 0xbf80, 0x4000                                                     # +2 in group 0, -2 in group 1
    add('cancellation_across_groups', canc2_a, canc2_b, np.zeros(8))
    add('acc_cancels_products', canc2_a * 0 + np.where(np.arange(16) == 0, 0x3f80, 0).astype(np.uint16), canc2_b * 0 + np.where(np.arange(16) == 0, 0x4000, 0).astype(np.uint16),
        np.full(8, -2.0, np.float32))
    # --- Inf / NaN
    k = 64
    inf_a = rnd_norm(k); inf_a[:, 3] = 0x7f80
    add('inf_operand', inf_a, rnd_norm(k), rng.standard_normal(k))
    ninf_a = rnd_norm(k); ninf_a[:, 3] = 0x7f80; ninf_a[:, 5] = 0xff80
    pb = rnd_norm(k) & 0x7fff                     # positive partners: +inf and -inf in one group
    add('inf_minus_inf', ninf_a, pb, np.zeros(k))
    ninf2 = rnd_norm(k); ninf2[:, 3] = 0x7f80; ninf2[:, 11] = 0xff80
    add('inf_minus_inf_across_groups', ninf2, pb, np.zeros(k))
    zinf_a = rnd_norm(k); zinf_a[:, 2] = 0x7f80
    zinf_b = rnd_norm(k); zinf_b[:, 2] = 0
    add('zero_times_inf', zinf_a, zinf_b, np.zeros(k))
    nan_a = rnd_norm(k); nan_a[:, 7] = 0x7fc0
    add('nan_operand', nan_a, rnd_norm(k), rng.standard_normal(k))
    add('inf_acc', rnd_norm(k), rnd_norm(k), np.full(k, np.inf, np.float32))
    add('nan_acc', rnd_norm(k), rnd_norm(k), np.full(k, np.nan, np.float32))
    infacc_a = rnd_norm(k); infacc_a[:, 0] = 0xff80
    add('inf_acc_minus_inf_product', infacc_a, pb, np.full(k, np.inf, np.float32))
    # --- overflow of the sum
    add('overflow_products', rnd_norm(k, 60, 64) & 0x7fff, rnd_norm(k, 60, 64) & 0x7fff, np.zeros(k))
    add('overflow_acc', rnd_norm(k, 50, 60) & 0x7fff, rnd_norm(k, 50, 60) & 0x7fff, np.full(k, 3e38, np.float32))
    # --- what the split really produces for tiny inputs: x ~ 2^-100 .. 2^-126 (third terms subnormal / zero in bf16)
    from oracle import oracle as orc
    x = (rng.standard_normal((n, 16)) * np.exp2(rng.integers(-126, -100, (n, 1)).astype(np.float64))).astype(np.float32)
    w = (rng.standard_normal((n, 16))).astype(np.float32)
    px, pw = orc.split3(x), orc.split3(w)
    for ta_, tb_ in ((2, 0), (1, 1), (0, 0)):
        add(f'split_of_tiny_x_term{ta_ + 1}{tb_ + 1}', px[ta_], pw[tb_], np.zeros(n))
    return np.array(fam), np.concatenate(A), np.concatenate(B), np.concatenate(C).astype(np.float32)


def oracle_result(a, b, c):
    from oracle import oracle as orc
    out = np.empty(len(c), np.float32)
    with np.errstate(all='ignore'):
        for i in range(len(c)):
            out[i] = orc.mfma_dot8(orc.mfma_dot8(c[i], a[i, :8], b[i, :8]), a[i, 8:], b[i, 8:])
    return out


def same_bits(x, y):
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    return (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))


def main():
    import torch
    from femasr_amd import _lib
    out_path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/mfma_bf16_corners.npz'
    fam, a, b, c = cases()
    lib = _lib.load()
    ta = torch.from_numpy(a.view(np.int16).copy()).cuda()
    tb = torch.from_numpy(b.view(np.int16).copy()).cuda()
    tc = torch.from_numpy(c.copy()).cuda()
    td = torch.empty(len(c), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_debug_mfma_bf16(None, _lib.ptr(ta), _lib.ptr(tb), _lib.ptr(tc), len(c), _lib.ptr(td)))
    torch.cuda.synchronize()
    d = td.cpu().numpy()
    o = oracle_result(a, b, c)
    ok = same_bits(d, o)
    print(f'{len(c)} cases, {int((~ok).sum())} differ from the oracle restatement')
    for f in dict.fromkeys(fam.tolist()):
        m = fam == f
        bad = m & ~ok
        line = f'  {f:36s} {int(m.sum()):5d} cases, {int(bad.sum()):5d} differ'
        if bad.any():
            i = int(np.nonzero(bad)[0][0])
            line += f'   e.g. case {i}: hw {d[i]!r} ({d[i:i+1].view(np.uint32)[0]:08x})  oracle {o[i]!r} ({o[i:i+1].view(np.uint32)[0]:08x})  acc {c[i]!r}'
        print(line)
    os.makedirs(os.path.dirname(out_path) or '.', exist_ok=True)
    np.savez_compressed(out_path, a=a, b=b, c=c, d_hw=d, d_oracle_at_probe_time=o, family=fam,
                        note=np.array('corner families of v_mfma_f32_32x32x16_bf16 run on an MI355X through femasr_debug_mfma_bf16 (tools/mfma_corner_probe.py); d_hw = the hardware'))


if __name__ == '__main__':
    main()
