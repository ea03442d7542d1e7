"""-m gpu, round 6:
  * the split GEMM with its A operand handed over as packed bf16 planes (femasr_conv_args.in_bf16s: pure LDS-DMA main loop) and with
    its output written as planes (out_bf16s: fc1 -> fc2) - the same bits as the fp32-row form and as the oracle;
  * femasr_layernorm_bf16s (norm1 / norm2 writing the planes qkv / fc1 read) against femasr_layernorm and the oracle's LayerNorm;
  * pack / unpack of activation rows against the oracle's three-term split;
  * the Swin chain LN -> fc1(+GELU, planes out) -> fc2(+res) through planes end to end against the oracle's ops."""
import ctypes

import numpy as np
import pytest
import torch

from femasr_amd import _lib

pytestmark = pytest.mark.gpu


def _planes_of(x):
    """Packed planes of fp32 rows (device tensor) -> (uint8 device buffer, rows, C)."""
    lib = _lib.load()
    rows, c = x.shape
    buf = torch.full((int(lib.femasr_packed_rows_bf16s_bytes(rows, c)),), 0xff, dtype=torch.uint8, device=x.device)
    _lib.check(lib.femasr_pack_rows_bf16s(None, _lib.ptr(x), rows, c, _lib.ptr(buf)))
    return buf


def _unpack(buf, rows, c):
    lib = _lib.load()
    out = torch.full((rows, c), float('nan'), dtype=torch.float32, device=buf.device)
    _lib.check(lib.femasr_unpack_rows_bf16s(None, _lib.ptr(buf), rows, c, _lib.ptr(out)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('rows,c', [(128, 256), (1000, 64), (129, 1024), (1, 16), (4096 + 17, 256)])
def test_pack_rows_bf16s_is_the_oracles_split(cuda_device, rows, c):
    """femasr_pack_rows_bf16s: x = x1 + x2 + x3 with the oracle's three bf16 terms (orc_split3), laid out
    [row block 128][step 16 ch][plane][granule 8 ch][row][8]; unpack is its exact inverse."""
    from oracle import oracle as orc
    rng = np.random.default_rng(rows + c)
    x = (rng.standard_normal((rows, c)) * np.exp2(rng.integers(-20, 20, (rows, 1)))).astype(np.float32)
    x[0, :4] = [0.0, -0.0, 1e-30, -3e38]          # zeros, a tiny value, near the top of the range
    xg = torch.from_numpy(x).to(cuda_device)
    buf = _planes_of(xg)
    assert np.array_equal(_unpack(buf, rows, c), x)          # (as VALUES: -0 = (-0) + 0 + 0 comes back as +0; the planes themselves are compared below)
    p = orc.split3(x)
    mb, ns = (rows + 127) // 128, c // 16
    got = buf.cpu().numpy().view(np.uint16).reshape(mb, ns, 3, 2, 128, 8)
    for pl in range(3):
        want = np.zeros((mb * 128, c), np.uint16)
        want[:rows] = p[pl]
        want = want.reshape(mb, 128, ns, 2, 8).transpose(0, 2, 3, 1, 4)
        assert np.array_equal(got[:, :, pl], want), f'plane {pl + 1}'


SHAPES = [('qkv', 1536, 256, 768, 0, 0), ('proj', 1000, 256, 256, 0, 1), ('fc1', 777, 256, 1024, 1, 0), ('fc2', 1300, 1024, 256, 0, 1),
          ('fc2_two_res', 515, 1024, 256, 0, 2), ('gelu_res', 640, 512, 512, 1, 1), ('ragged', 333, 64, 200, 1, 1), ('one_block_tail', 129, 128, 36, 0, 1),
          ('narrow', 64, 64, 3, 0, 0), ('k64_steps4', 4096 + 100, 64, 128, 0, 0)]


@pytest.mark.parametrize('name,rows,cin,cout,act,nres', SHAPES, ids=[s[0] for s in SHAPES])
def test_linear_bf16s_planes_in_bit_exact(cuda_device, name, rows, cin, cout, act, nres):
    """in_bf16s: A by LDS-DMA from the packed planes == the fp32-row form == orc_linear_bf16s, every epilogue, ragged M / N."""
    import gpu_utils as G
    from oracle import oracle as orc
    rng = np.random.default_rng(len(name) * 7 + rows)
    x = (rng.standard_normal((rows, cin)) * 1.5).astype(np.float32)
    w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    r1 = rng.standard_normal((rows, cout)).astype(np.float32) if nres >= 1 else None
    r2 = rng.standard_normal((rows, cout)).astype(np.float32) if nres >= 2 else None
    yo = orc.linear_bf16s(x, w, b, act, r1, r2)
    w_khwc = np.ascontiguousarray(w.T).reshape(1, 1, cin, cout)
    kw = dict(act=act, res1=None if r1 is None else r1.reshape(1, rows, 1, cout), res2=None if r2 is None else r2.reshape(1, rows, 1, cout), bf16s=True)
    y = G.conv2d(x.reshape(1, rows, 1, cin), w_khwc, b, 1, planes_in=True, **kw).reshape(rows, cout)
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), f'{name}: {(y != yo).sum()} of {y.size} differ, max-abs {np.nanmax(np.abs(y - yo)):.3e}'
    y0 = G.conv2d(x.reshape(1, rows, 1, cin), w_khwc, b, 1, **kw).reshape(rows, cout)
    assert np.array_equal(y.view(np.uint32), y0.view(np.uint32))


@pytest.mark.parametrize('name,rows,cin,cout,act', [('fc1', 777, 256, 1024, 1), ('plain', 1300, 1024, 256, 0), ('ragged', 333, 64, 208, 1), ('tail', 129, 128, 48, 0)])
def test_linear_bf16s_planes_out_bit_exact(cuda_device, name, rows, cin, cout, act):
    """out_bf16s: the epilogue writes the three planes of act(x W + b) in the packed layout; unpacked they are the oracle's fp32 result."""
    import gpu_utils as G
    from oracle import oracle as orc
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, cin)) * 1.5).astype(np.float32)
    w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    yo = orc.linear_bf16s(x, w, b, act)
    y = G.conv2d(x.reshape(1, rows, 1, cin), np.ascontiguousarray(w.T).reshape(1, 1, cin, cout), b, 1, act=act, bf16s=True, planes_in=True,
                 planes_out=True).reshape(rows, cout)
    assert np.array_equal(y, yo), f'{name}: {(y != yo).sum()} of {y.size} differ'          # (values: a -0 output unpacks as +0)


def test_linear_bf16s_planes_refusals(cuda_device):
    import gpu_utils as G
    x, w, b = np.zeros((1, 8, 1, 64), np.float32), np.zeros((1, 1, 64, 16), np.float32), np.zeros(16, np.float32)
    with pytest.raises(_lib.FemasrError):          # planes without the split arithmetic
        G.conv2d(x, w, b, 1, planes_in=True)
    with pytest.raises(_lib.FemasrError):          # plane output with a residual operand
        G.conv2d(x, w, b, 1, bf16s=True, planes_in=True, planes_out=True, res1=np.zeros((1, 8, 1, 16), np.float32))
    with pytest.raises(_lib.FemasrError):          # plane output from fp32 rows (not instantiated: the network never needs it)
        G.conv2d(x, w, b, 1, bf16s=True, planes_out=True)
    with pytest.raises(_lib.FemasrError):          # Cout % 16 != 0
        G.conv2d(x, np.zeros((1, 1, 64, 8), np.float32), np.zeros(8, np.float32), 1, bf16s=True, planes_in=True, planes_out=True)


@pytest.mark.parametrize('rows', [32, 33, 1000, 128 * 5, 4096 + 17])
def test_layernorm_bf16s_bit_exact(cuda_device, rows):
    """femasr_layernorm_bf16s == femasr_layernorm == the oracle, as planes (network_swinir.py:243,277)."""
    import gpu_utils as G
    from oracle import oracle as orc
    lib = _lib.load()
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, 256)) * 3 + rng.standard_normal((rows, 1)) * 5).astype(np.float32)
    g = (1 + 0.2 * rng.standard_normal(256)).astype(np.float32)
    b = (0.3 * rng.standard_normal(256)).astype(np.float32)
    ref = orc.layernorm(x, g, b)
    assert np.array_equal(G.layernorm(x, g, b).view(np.uint32), ref.view(np.uint32))
    tx, tg, tb = (torch.from_numpy(a).to(cuda_device) for a in (x, g, b))
    buf = torch.full((int(lib.femasr_packed_rows_bf16s_bytes(rows, 256)),), 0xff, dtype=torch.uint8, device=cuda_device)
    _lib.check(lib.femasr_layernorm_bf16s(None, _lib.ptr(tx), rows, 256, _lib.ptr(tg), _lib.ptr(tb), ctypes.c_float(1e-5), _lib.ptr(buf)))
    got = _unpack(buf, rows, 256)
    assert np.array_equal(got, ref), f'{(got != ref).sum()} of {ref.size} differ'
    # the planes themselves are the pack of the fp32 result (same record layout)
    want = _planes_of(torch.from_numpy(ref).to(cuda_device)).cpu().numpy().view(np.uint16).reshape(-1, 16, 3, 2, 128, 8)
    have = buf.cpu().numpy().view(np.uint16).reshape(-1, 16, 3, 2, 128, 8)
    full = rows // 128
    assert np.array_equal(have[:full], want[:full])
    if rows % 128:
        assert np.array_equal(have[full, :, :, :, :rows % 128], want[full, :, :, :, :rows % 128])
    with pytest.raises(_lib.FemasrError):
        _lib.check(lib.femasr_layernorm_bf16s(None, _lib.ptr(tx), rows, 128, _lib.ptr(tg), _lib.ptr(tb), ctypes.c_float(1e-5), _lib.ptr(buf)))


def test_swin_mlp_chain_through_planes(cuda_device):
    """norm2 -> fc1 (+GELU) -> fc2 (+residual) with every hand-over as packed planes (what FeMaSRNet's forward launches in linear_math
    'bf16_split') == the oracle's three ops on fp32 rows (network_swinir.py:14-30,276-277)."""
    from oracle import oracle as orc
    lib = _lib.load()
    rows, C, Hd = 2000, 256, 1024
    rng = np.random.default_rng(6)
    x = rng.standard_normal((rows, C)).astype(np.float32)
    g, be = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32), (0.1 * rng.standard_normal(C)).astype(np.float32)
    w1, b1 = (rng.standard_normal((Hd, C)) / 16).astype(np.float32), (0.1 * rng.standard_normal(Hd)).astype(np.float32)
    w2, b2 = (rng.standard_normal((C, Hd)) / 32).astype(np.float32), (0.1 * rng.standard_normal(C)).astype(np.float32)
    n = orc.layernorm(x, g, be)
    hdn = orc.linear_bf16s(n, w1, b1, 1)
    yo = orc.linear_bf16s(hdn, w2, b2, 0, x)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device)
    tx, tg, tbe, tw1, tb1, tw2, tb2 = map(d, (x, g, be, w1, b1, w2, b2))

    def packw(tw, o, i):
        buf = torch.empty(int(lib.femasr_packed_weight_bf16s_bytes(o, i)), dtype=torch.uint8, device=cuda_device)
        _lib.check(lib.femasr_repack_k1_bf16s(None, _lib.ptr(tw), o, i, _lib.ptr(buf)))
        return buf
    p1, p2 = packw(tw1, Hd, C), packw(tw2, C, Hd)
    pn = torch.empty(int(lib.femasr_packed_rows_bf16s_bytes(rows, C)), dtype=torch.uint8, device=cuda_device)
    ph = torch.empty(int(lib.femasr_packed_rows_bf16s_bytes(rows, Hd)), dtype=torch.uint8, device=cuda_device)
    out = torch.full((rows, C), float('nan'), dtype=torch.float32, device=cuda_device)
    _lib.check(lib.femasr_layernorm_bf16s(None, _lib.ptr(tx), rows, C, _lib.ptr(tg), _lib.ptr(tbe), ctypes.c_float(1e-5), _lib.ptr(pn)))

    def lin(in_planes, cin, cout, wp, bias, act=0, res=None, out_planes=None, out_rows=None):
        a = _lib.ConvArgs()
        a.in_, a.in_bf16s = None, in_planes.data_ptr()
        a.B, a.H, a.W, a.Cin, a.Cout, a.ksz, a.stride, a.pad = 1, rows, 1, cin, cout, 1, 1, 0
        a.Ho, a.Wo, a.act = rows, 1, act
        a.bias, a.w_bf16s = bias.data_ptr(), wp.data_ptr()
        a.res1 = None if res is None else res.data_ptr()
        a.out = None if out_rows is None else out_rows.data_ptr()
        a.out_bf16s = None if out_planes is None else out_planes.data_ptr()
        _lib.check(lib.femasr_conv2d(None, ctypes.byref(a)))
    lin(pn, C, Hd, p1, tb1, act=_lib.ACT_GELU, out_planes=ph)
    lin(ph, Hd, C, p2, tb2, res=tx, out_rows=out)
    torch.cuda.synchronize()
    assert np.array_equal(_unpack(ph, rows, Hd), hdn)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), yo.view(np.uint32))


def test_test_tile_u8_equals_the_fp32_tile_path(cuda_device):
    """FeMaSRNet.test_tile_u8 (uint8 crops -> test_u8 -> uint8 paste; inference_femasr.py:58-67 with femasr_arch.py:387-447) ==
    output_to_u8(test_tile(u8_to_input(img))) bit for bit: ragged border tiles, several shape classes, batch 2, BGR order, and
    `test_u8(out=...)` writing into a slice."""
    import gpu_utils as G
    from femasr_amd import imgproc, synth
    from helpers import synth_weights
    w = synth_weights('x4', 11, 'trained')
    net = G.build_net('x4', w, cuda_device)
    rng = np.random.default_rng(5)
    img = torch.from_numpy(rng.integers(0, 256, (2, 75, 100, 3), dtype=np.uint8)).to(cuda_device)
    want = torch.stack([imgproc.output_to_u8(net.test_tile(imgproc.u8_to_input(img[i]), 32, 8)) for i in range(2)])
    got = net.test_tile_u8(img, 32, 8)
    assert got.dtype == torch.uint8 and got.shape == (2, 300, 400, 3) and torch.equal(got, want), int((got != want).sum())
    one = net.test_tile_u8(img[1], 32, 8)                               # (H,W,3) in -> (sH,sW,3) out
    assert torch.equal(one, want[1])
    bgr = net.test_tile_u8(img.flip(-1), 32, 8, bgr=True)               # cv2-style channel order in and out
    assert torch.equal(bgr.flip(-1), want)
    buf = torch.zeros((3, 160, 192, 3), dtype=torch.uint8, device=cuda_device)
    y = net.test_u8(img[:1, :40, :48], out=buf[1:2])
    assert y.data_ptr() == buf[1:2].data_ptr() and torch.equal(buf[1:2], net.test_u8(img[:1, :40, :48])) and int(buf[0].max()) == 0 and int(buf[2].max()) == 0
    with pytest.raises(ValueError):
        net.test_u8(img[:1, :40, :48], out=buf[1:2, :, :100])
