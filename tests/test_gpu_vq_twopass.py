"""Two-pass exact codebook search (kernels_vq.hip) against the oracle and the single-pass fp32 path: bit-exact indices
and z_q in every regime, including the ones built to stress the candidate bound (exact ties, near ties closer than the
bf16 resolution, tiny codebooks against large |z|, clusters that overflow the candidate list)."""
import numpy as np
import pytest

from femasr_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _case(name):
    if name == 'trained':
        cb = synth.uniform(21, 'tp.cb', (1024, 512), -1.0, 1.0)
        cb[700] = cb[13]; cb[14] = cb[13]; cb[901] = cb[13]
        z = synth.uniform(22, 'tp.z', (1000, 512), -1.0, 1.0)
        z[0] = cb[13] * 0.97
        z[5] = cb[901]
        z[7] = 0.0
    elif name == 'init':                    # reference init: |e|^2 ~ 1.6e-4 against |z|^2 ~ 170 -> distances quantised, many ties
        cb = synth.uniform(23, 'tp.cb', (1024, 512), -1.0 / 1024, 1.0 / 1024)
        z = synth.uniform(24, 'tp.z', (777, 512), -1.0, 1.0)
    elif name == 'near_ties':               # pairs of codes 1 fp32 ulp apart in one coordinate: far below bf16 resolution
        cb = synth.uniform(25, 'tp.cb', (512, 256), -1.0, 1.0)
        for j in range(0, 512, 2):
            cb[j + 1] = cb[j]
            cb[j + 1, j % 256] = np.nextafter(cb[j, j % 256], np.float32(2.0))
        z = synth.uniform(26, 'tp.z', (650, 256), -1.0, 1.0)
        z[:256] = cb[::2] + synth.uniform(27, 'tp.n', (256, 256), -1e-3, 1e-3)
    elif name == 'cluster_overflow':        # 50 codes within 1e-6 of each other around the rows: more than 32 survivors
        cb = synth.uniform(28, 'tp.cb', (256, 128), -1.0, 1.0)
        cb[100:150] = cb[100] + synth.uniform(29, 'tp.d', (50, 128), -1e-6, 1e-6)
        z = synth.uniform(30, 'tp.z', (300, 128), -1.0, 1.0)
        z[::3] = cb[100] + synth.uniform(31, 'tp.e', (100, 128), -1e-4, 1e-4)
    elif name == 'large_z':                 # activations far larger than the codes
        cb = synth.uniform(32, 'tp.cb', (128, 64), -0.05, 0.05)
        z = synth.uniform(33, 'tp.z', (129, 64), -30.0, 30.0)
    elif name == 'identical_codes':         # every code the same: all rows take the "every code" path, answer 0
        cb = np.tile(synth.uniform(34, 'tp.cb', (1, 64), -1.0, 1.0), (64, 1))
        z = synth.uniform(35, 'tp.z', (70, 64), -1.0, 1.0)
    else:
        raise KeyError(name)
    return z.astype(np.float32), cb.astype(np.float32)


CASES = ['trained', 'init', 'near_ties', 'cluster_overflow', 'large_z', 'identical_codes']


@pytest.mark.parametrize('name', CASES)
def test_twopass_bit_exact(cuda_device, name):
    import gpu_utils as G
    z, cb = _case(name)
    idx_ref, zq_ref = orc.vq(z, cb)
    idx, zq, _, _ = G.vq(z, cb, mode='twopass')
    assert np.array_equal(idx, idx_ref), f'{int((idx != idx_ref).sum())} of {idx.size} indices differ'
    assert np.array_equal(zq.view(np.uint32), zq_ref.view(np.uint32))
    if cb.shape[0] % 128 == 0:
        idx1, zq1, _, _ = G.vq(z, cb, mode='gemm')
        assert np.array_equal(idx, idx1) and np.array_equal(zq.view(np.uint32), zq1.view(np.uint32))
    if name == 'identical_codes':
        assert (idx == 0).all()


@pytest.mark.parametrize('name', CASES)
def test_candidates_contain_the_exact_first_min(cuda_device, name):
    import gpu_utils as G
    z, cb = _case(name)
    idx_ref, _ = orc.vq(z, cb)
    cand, cnt = G.vq_candidates(z, cb)
    every = cnt == 0xFFFF
    n = np.where(every, 0, cnt).astype(np.int64)
    assert (n <= 32).all() and ((n >= 1) | every).all()
    hit = every.copy()
    for k in range(32):
        hit |= (k < n) & (cand[:, k].astype(np.int64) == idx_ref)
    assert hit.all(), f'{int((~hit).sum())} rows lost their first-min'
    print(f'{name}: mean candidates {n[~every].mean() if (~every).any() else 0:.2f}, max {n.max()}, '
          f'"every code" rows {int(every.sum())} of {len(cnt)}')
    if name == 'cluster_overflow':
        assert every.any()
    if name in ('trained', 'init'):
        assert not every.any() and n.mean() < 6


@pytest.mark.parametrize('seed', range(6))
def test_twopass_fuzz_vs_single_pass(cuda_device, seed):
    """Larger random problems, both codebook scales: two-pass == single-pass (which the other tests pin to the oracle)."""
    import gpu_utils as G
    rng = np.random.RandomState(900 + seed)
    m = int(rng.randint(1, 9000))
    n_e = int(rng.choice([128, 256, 512, 1024]))
    d = int(rng.choice([64, 128, 256, 512]))
    scale = float(rng.choice([1.0, 1.0 / n_e, 0.1]))
    cb = synth.uniform(910 + seed, 'f.cb', (n_e, d), -scale, scale)
    z = synth.uniform(920 + seed, 'f.z', (m, d), -1.0, 1.0) * np.float32(rng.choice([1.0, 0.05, 4.0]))
    for t in range(min(m, 16)):
        j = int(rng.randint(0, n_e))
        z[t] = cb[j]
        cb[(j * 7 + 3) % n_e] = cb[j]
    idx1, zq1, _, _ = G.vq(z, cb, mode='gemm')
    idx2, zq2, _, _ = G.vq(z, cb, mode='twopass')
    assert np.array_equal(idx1, idx2), f'{int((idx1 != idx2).sum())} of {m} indices differ (n_e={n_e}, d={d}, scale={scale})'
    assert np.array_equal(zq1.view(np.uint32), zq2.view(np.uint32))


def test_twopass_shape_rule_and_model_default(cuda_device):
    from femasr_amd import _lib
    lib = _lib.load()
    assert lib.femasr_vq_twopass_ok(1024, 512) and lib.femasr_vq_twopass_ok(64, 64)
    assert not lib.femasr_vq_twopass_ok(2048, 512) and not lib.femasr_vq_twopass_ok(1024, 96) and not lib.femasr_vq_twopass_ok(1024, 32)
    assert lib.femasr_vq_aux_bytes(1024, 512) == 1024 * 512 * 2 + 1026 * 4
