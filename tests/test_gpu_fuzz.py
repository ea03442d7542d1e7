"""Randomised (fixed-seed) shape sweep of the conv kernels against the oracle: every tiling / edge-tile / epilogue
branch (full and ragged tiles, Cout not a multiple of the tile, odd sizes, fused x2, residual count 0/1/2, LN / GN
prologues, GELU) with small tensors the CPU oracle finishes in milliseconds.  fp32 paths must be bit-exact; the
bf16x3 path is bounded at 1e-4 of the output scale (network-level contract: 1e-3)."""
import os

import numpy as np
import pytest

from femasr_amd import _lib, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

_MULT = int(os.environ.get('FEMASR_FUZZ_MULT', '1'))     # FEMASR_FUZZ_MULT=8 for a longer one-off sweep


def _cases(seed, n):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        out.append(dict(id=i, b=int(rng.randint(1, 3)), h=int(rng.randint(3, 37)), w=int(rng.randint(3, 41)),
                        cin=int(rng.choice([32, 64, 96, 128, 256])), cout=int(rng.choice([3, 24, 32, 64, 96, 128, 160, 256, 320])),
                        up2=bool(rng.randint(0, 2)), gn=bool(rng.randint(0, 2)), nres=int(rng.randint(0, 3))))
    return out


def _tensors(c, tag, ho, wo):
    x = synth.uniform(100 + c['id'], tag + 'x', (c['b'], c['h'], c['w'], c['cin']), -2.0, 3.0)
    w = synth.uniform(100 + c['id'], tag + 'w', (3, 3, c['cin'], c['cout']), -0.1, 0.1)
    bias = synth.uniform(100 + c['id'], tag + 'b', (c['cout'],), -0.5, 0.5)
    res = [synth.uniform(100 + c['id'], tag + f'r{k}', (c['b'], ho, wo, c['cout']), -1, 1) for k in range(c['nres'])]
    return x, w, bias, (res + [None, None])[:2]


@pytest.mark.parametrize('c', _cases(2024, 36 * _MULT), ids=lambda c: 'f%(id)d_%(b)dx%(h)dx%(w)d_%(cin)d_%(cout)d_u%(up2)d_g%(gn)d_r%(nres)d' % c)
def test_fuzz_conv3x3_fp32_bit_exact(cuda_device, c):
    import gpu_utils as G
    up = c['up2'] and not c['gn']            # the fused x2 has no GN prologue (never needed by the network)
    ho, wo = (2 * c['h'], 2 * c['w']) if up else (c['h'], c['w'])
    x, w, bias, (r1, r2) = _tensors(c, 'fz', ho, wo)
    if c['gn'] and c['cin'] % 32 == 0:
        gamma = synth.uniform(7, 'fzg', (c['cin'],), 0.5, 1.5)
        beta = synth.uniform(7, 'fzbe', (c['cin'],), -0.5, 0.5)
        a_, b_ = orc.gn_coeffs(x, gamma, beta)
        ref = orc.conv2d(orc.scale_shift_silu(x, a_, b_), w, bias, 3, 1, 1, up, res1=r1, res2=r2)
        got = G.conv2d(x, w, bias, 3, 1, 1, up, prologue=_lib.PRO_GN_SILU, pro=(a_, b_, None), res1=r1, res2=r2)
    else:
        ref = orc.conv2d(x, w, bias, 3, 1, 1, up, res1=r1, res2=r2)
        got = G.conv2d(x, w, bias, 3, 1, 1, up, res1=r1, res2=r2)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), \
        f'max-abs {np.abs(got - ref).max():.3e}'


@pytest.mark.parametrize('c', _cases(77, 30 * _MULT), ids=lambda c: 'b%(id)d_%(b)dx%(h)dx%(w)d_%(cin)d_%(cout)d_u%(up2)d_g%(gn)d_r%(nres)d' % c)
def test_fuzz_conv3x3_bf16x3_within_tolerance(cuda_device, c):
    import gpu_utils as G
    import torch
    up = c['up2'] and not c['gn']
    ho, wo = (2 * c['h'], 2 * c['w']) if up else (c['h'], c['w'])
    x, w, bias, (r1, r2) = _tensors(c, 'bz', ho, wo)
    want_part = (c['cout'] % 32 == 0) and (c['cout'] // 32) in (1, 2, 4, 8)     # channels per group must be a power of two
    kw = dict(res1=r1, res2=r2, bf16x3=True, gn_part=want_part)
    if c['gn']:
        gamma = synth.uniform(8, 'bzg', (c['cin'],), 0.5, 1.5)
        beta = synth.uniform(8, 'bzbe', (c['cin'],), -0.5, 0.5)
        a_, b_ = orc.gn_coeffs(x, gamma, beta)
        ref = orc.conv2d(orc.scale_shift_silu(x, a_, b_), w, bias, 3, 1, 1, up, res1=r1, res2=r2)
        got = G.conv2d(x, w, bias, 3, 1, 1, up, prologue=_lib.PRO_GN_SILU, pro=(a_, b_, None), **kw)
    else:
        ref = orc.conv2d(x, w, bias, 3, 1, 1, up, res1=r1, res2=r2)
        got = G.conv2d(x, w, bias, 3, 1, 1, up, **kw)
    part = None
    if want_part:
        got, part = got
    scale = max(1.0, float(np.abs(ref).max()))
    assert got.shape == ref.shape and float(np.abs(got - ref).max()) <= 1e-4 * scale
    if part is not None:                      # fused GroupNorm partial moments of the output
        assert not torch.isnan(part).any()
        g2 = synth.uniform(9, 'bzg2', (c['cout'],), 0.5, 1.5)
        b2 = synth.uniform(9, 'bzb2', (c['cout'],), -0.5, 0.5)
        a, bb = G.gn_coeffs_from_partials(part, got.shape[1], got.shape[2], c['cout'], g2, b2)
        a_ref, b_ref = orc.gn_coeffs(got, g2, b2)
        assert np.abs(a - a_ref).max() <= 1e-5 * np.abs(a_ref).max()
        assert np.abs(bb - b_ref).max() <= 2e-5 * max(1.0, np.abs(b_ref).max())


def _lin_cases(seed, n):
    rng = np.random.RandomState(seed)
    return [dict(id=i, rows=int(rng.randint(1, 700)), cin=int(rng.choice([32, 64, 256, 512, 1024])),
                 cout=int(rng.choice([3, 32, 64, 96, 128, 256, 384, 768])), ln=bool(rng.randint(0, 2)),
                 act=int(rng.randint(0, 2)), nres=int(rng.randint(0, 3))) for i in range(n)]


@pytest.mark.parametrize('c', _lin_cases(5, 30 * _MULT), ids=lambda c: 'l%(id)d_%(rows)d_%(cin)d_%(cout)d_ln%(ln)d_a%(act)d_r%(nres)d' % c)
def test_fuzz_linear_bit_exact(cuda_device, c):
    """1x1 conv / nn.Linear path (LDS-DMA GEMM): ragged row tiles, all epilogue variants (scalar and float4 stores)."""
    import gpu_utils as G
    rows, cin, cout = c['rows'], c['cin'], c['cout']
    ln = c['ln'] and cin == 256                         # ln_stats is built for C = 256 (the Swin width)
    x = synth.uniform(200 + c['id'], 'lzx', (rows, cin), -3, 4)
    w = synth.uniform(200 + c['id'], 'lzw', (cin, cout), -0.1, 0.1)
    bias = synth.uniform(200 + c['id'], 'lzb', (cout,), -0.5, 0.5)
    res = [synth.uniform(200 + c['id'], f'lzr{k}', (rows, cout), -1, 1) for k in range(c['nres'])]
    r1, r2 = (res + [None, None])[:2]
    xin, xg = x, x
    if ln:
        gamma = synth.uniform(6, 'lzg', (256,), 0.5, 1.5)
        beta = synth.uniform(6, 'lzbe', (256,), -0.5, 0.5)
        xin = orc.layernorm(x, gamma, beta)
        xg = G.layernorm(x, gamma, beta)
        assert np.array_equal(xg.view(np.uint32), xin.view(np.uint32))
    ref = orc.linear(xin, w, bias, act=c['act'], res=r1)
    if r2 is not None:
        ref = ref + r2                                  # second residual: one more fp32 add, in this order
    got = G.conv2d(xg.reshape(1, rows, 1, cin), w.reshape(1, 1, cin, cout), bias, 1, act=c['act'],
                   res1=None if r1 is None else r1.reshape(1, rows, 1, cout),
                   res2=None if r2 is None else r2.reshape(1, rows, 1, cout)).reshape(rows, cout)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f'max-abs {np.abs(got - ref).max():.3e}'


def _gen_cases(seed, n):
    rng = np.random.RandomState(seed)
    return [dict(id=i, b=int(rng.randint(1, 3)), h=int(rng.randint(4, 34)), w=int(rng.randint(4, 34)),
                 cin=int(rng.choice([3, 5, 32, 64, 96])), cout=int(rng.choice([3, 32, 64, 96, 128, 256])),
                 k=int(rng.choice([1, 3, 4])), s=int(rng.choice([1, 2])), p=int(rng.randint(0, 2)),
                 nres=int(rng.randint(0, 3)), act=int(rng.randint(0, 2))) for i in range(n)]


@pytest.mark.parametrize('c', _gen_cases(31, 24 * _MULT), ids=lambda c: 'g%(id)d_%(b)dx%(h)dx%(w)d_%(cin)d_%(cout)d_k%(k)ds%(s)dp%(p)d_r%(nres)d_a%(act)d' % c)
def test_fuzz_general_conv_bit_exact(cuda_device, c):
    """implicit-GEMM path: in_conv-like (Cin not a multiple of 32), strided and unpadded convs, k = 1 / 3 / 4."""
    import gpu_utils as G
    if c['h'] + 2 * c['p'] < c['k'] or c['w'] + 2 * c['p'] < c['k']:
        pytest.skip('kernel larger than the padded image')
    ho, wo = (c['h'] + 2 * c['p'] - c['k']) // c['s'] + 1, (c['w'] + 2 * c['p'] - c['k']) // c['s'] + 1
    x = synth.uniform(300 + c['id'], 'gzx', (c['b'], c['h'], c['w'], c['cin']), -2, 2)
    w = synth.uniform(300 + c['id'], 'gzw', (c['k'], c['k'], c['cin'], c['cout']), -0.2, 0.2)
    bias = synth.uniform(300 + c['id'], 'gzb', (c['cout'],), -0.5, 0.5)
    res = [synth.uniform(300 + c['id'], f'gzr{k}', (c['b'], ho, wo, c['cout']), -1, 1) for k in range(c['nres'])]
    r1, r2 = (res + [None, None])[:2]
    ref = orc.conv2d(x, w, bias, c['k'], c['s'], c['p'], False, act=c['act'], res1=r1, res2=r2)
    got = G.conv2d(x, w, bias, c['k'], c['s'], c['p'], False, act=c['act'], res1=r1, res2=r2)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), \
        f'max-abs {np.abs(got - ref).max():.3e}'


@pytest.mark.parametrize('seed', range(6 * _MULT))
def test_fuzz_window_attention_bit_exact(cuda_device, seed):
    import gpu_utils as G
    rng = np.random.RandomState(900 + seed)
    b, h, w = int(rng.randint(1, 3)), 8 * int(rng.randint(1, 5)), 8 * int(rng.randint(1, 5))
    heads = int(rng.choice([1, 2, 8]))
    c = 32 * heads
    shift = int(rng.choice([0, 4])) if min(h, w) > 8 else 0
    qkv = synth.uniform(400 + seed, 'aq', (b, h * w, 3 * c), -3, 3)
    table = synth.uniform(400 + seed, 'at', (225, heads), -1, 1)
    got = G.window_attention(qkv, b, h, w, c, heads, shift, table)
    ref = orc.window_attention(qkv, b, h, w, c, heads, shift, table)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f'max-abs {np.abs(got - ref).max():.3e}'


@pytest.mark.parametrize('seed', range(5 * _MULT))
def test_fuzz_vq_bit_exact(cuda_device, seed):
    import gpu_utils as G
    rng = np.random.RandomState(700 + seed)
    m, n_e, d = int(rng.randint(1, 600)), int(rng.choice([128, 256, 1024])), int(rng.choice([32, 64, 256, 512]))
    cb = synth.uniform(500 + seed, 'vc', (n_e, d), -1, 1)
    z = synth.uniform(500 + seed, 'vz', (m, d), -1, 1)
    for t in range(min(m, 8)):                        # a few exact and near ties
        j = int(rng.randint(0, n_e))
        z[t] = cb[j]
        cb[(j * 7 + 3) % n_e] = cb[j]
    idx_ref, zq_ref = orc.vq(z, cb)
    idx, zq, _, _ = G.vq(z, cb)
    assert np.array_equal(idx, idx_ref)
    assert np.array_equal(zq.view(np.uint32), zq_ref.view(np.uint32))


def _net_cases(seed, n):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        cfg = str(rng.choice(['x4', 'x2', 'hq']))
        if cfg == 'hq':                              # forward() of the HQ net: multiples of 8, no padding
            h, w = 8 * int(rng.randint(2, 8)), 8 * int(rng.randint(2, 8))
        else:                                        # test(): any size the mirror pad accepts
            h, w = int(rng.randint(16, 57)), int(rng.randint(16, 57))
        out.append(dict(id=i, cfg=cfg, b=int(rng.randint(1, 4)), h=h, w=w, streams=int(rng.choice([1, 2, 3])),
                        math=str(rng.choice(['fp32_strict', 'fp32', 'bf16x3']))))
    return out


@pytest.mark.parametrize('c', _net_cases(4242, 6 * _MULT), ids=lambda c: 'n%(id)d_%(cfg)s_%(b)dx%(h)dx%(w)d_s%(streams)d_%(math)s' % c)
def test_fuzz_network_vs_oracle(cuda_device, c):
    """Whole network at random small sizes / batch / stream count / math mode: fp32_strict bit-exact against the oracle
    (output and VQ indices); the default fp32 mode (hardware exp2 / rcp SiLU in the Winograd convs) identical indices and
    <= 1e-4; bf16x3 identical indices and <= 1e-3 (workspace planner, stream fork/join, ragged tiles in every layer)."""
    import gpu_utils as G
    import torch
    from helpers import oracle_net, synth_weights
    w = synth_weights(c['cfg'], 21 + c['id'], 'trained')
    net = G.build_net(c['cfg'], w, cuda_device)
    net.num_streams, net.decoder_math = c['streams'], c['math']
    x = synth.synth_input(50 + c['id'], (c['b'], 3, c['h'], c['w']))
    xt = torch.from_numpy(x).to(cuda_device)
    onet = oracle_net(c['cfg'], w)
    if c['cfg'] == 'hq':
        y, _, _, idx_list = net(xt)
        y, idx = y.cpu().numpy(), idx_list[0].cpu().numpy()
        yo, io = onet.forward(x)
    else:
        y, idx = net.test_with_indices(xt)
        y, idx = y.cpu().numpy(), idx.cpu().numpy()
        yo, io = onet.test(x, return_indices=True)
    assert np.array_equal(idx.reshape(-1), np.asarray(io).reshape(-1)), 'VQ indices differ from the oracle'
    if c['math'] == 'fp32_strict':
        assert np.array_equal(y, yo), f'not bit-identical: max-abs {np.abs(y - yo).max():.3e}'
    else:
        assert float(np.abs(y - yo).max()) < (1e-4 if c['math'] == 'fp32' else 1e-3)


@pytest.mark.parametrize('seed', range(8 * _MULT))
def test_fuzz_gn_and_ln_moments_bit_exact(cuda_device, seed):
    import gpu_utils as G
    rng = np.random.RandomState(1200 + seed)
    b, h, w = int(rng.randint(1, 4)), int(rng.randint(1, 70)), int(rng.randint(1, 50))
    c = int(rng.choice([32, 64, 96, 128, 160, 256, 512]))
    off = float(rng.choice([0.0, 10.0, -50.0]))              # cancellation-prone offsets
    x = synth.uniform(600 + seed, 'mx', (b, h, w, c), off - 2, off + 3)
    gamma = synth.uniform(600 + seed, 'mg', (c,), 0.5, 1.5)
    beta = synth.uniform(600 + seed, 'mb', (c,), -0.5, 0.5)
    a, bb = G.gn_coeffs(x, gamma, beta)
    a_ref, b_ref = orc.gn_coeffs(x, gamma, beta)
    assert np.array_equal(a.view(np.uint32), a_ref.view(np.uint32)) and np.array_equal(bb.view(np.uint32), b_ref.view(np.uint32))
    rows = int(rng.randint(1, 3000))
    xl = synth.uniform(700 + seed, 'lx', (rows, 256), off - 4, off + 6)
    g2 = synth.uniform(700 + seed, 'lg', (256,), 0.5, 1.5)
    b2 = synth.uniform(700 + seed, 'lb', (256,), -0.5, 0.5)
    got = G.layernorm(xl, g2, b2)
    ref = orc.layernorm(xl, g2, b2)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize('seed', range(2 * _MULT))       # (round 6: 2 cases instead of 4 by default - each costs ~25 s of CPU oracle; FEMASR_FUZZ_MULT widens)
def test_fuzz_test_tile_vs_oracle(cuda_device, seed):
    """`test_tile` with random tile size / pad (ragged border tiles, several shape classes) == the oracle's test_tile.  The first case of
    every two runs the product default arithmetic of the linears, the other the fp32 chain on both sides (this test is about the tiling
    logic; the restated matrix-instruction arithmetic costs ~30x on the CPU side and has its own tests)."""
    import gpu_utils as G
    import torch
    from helpers import oracle_net, synth_weights
    rng = np.random.RandomState(1500 + seed)
    w = synth_weights('x4', 31 + seed, 'trained')
    net = G.build_net('x4', w, cuda_device)
    lm = 'bf16_split' if seed % 2 == 0 else 'fp32'
    net.linear_math = lm
    net.num_streams = int(rng.choice([1, 2]))
    ts, pad = int(rng.choice([24, 32, 40, 48])), int(rng.choice([0, 4, 8]))

    def size():          # whole tiles plus a ragged remainder the mirror pad of test() accepts (>= 12 px; the reference
        rem = int(rng.choice([0, int(rng.randint(12, ts))]))      # also rejects narrower tiles)
        return ts * int(rng.randint(1, 3)) + rem
    h, wd = size(), size()
    x = synth.synth_input(60 + seed, (1, 3, h, wd))
    y = net.test_tile(torch.from_numpy(x).to(cuda_device), ts, pad).cpu().numpy()
    yo = oracle_net('x4', w, linear_math=lm).test_tile(x, ts, pad)
    assert y.shape == yo.shape == (1, 3, 4 * h, 4 * wd)
    assert np.array_equal(y, yo), f'max-abs {np.abs(y - yo).max():.3e}'
