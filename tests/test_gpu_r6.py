"""-m gpu, round 6:
  * the uint8 tile path (FeMaSRNet.test_tile_u8: crops, forwards, the all-gather and the paste on bytes);
  * the corners of v_mfma_f32_32x32x16_bf16 on THIS box against the committed corner fixture and the oracle's restatement, and the split
    GEMM on inputs whose split terms are bf16 subnormals."""
import numpy as np
import pytest
import torch

from femasr_amd import _lib

pytestmark = pytest.mark.gpu


def test_mfma_bf16_corner_families_vs_fixture_and_oracle(cuda_device):
    """tests/golden/mfma_bf16_corners.npz (recorded on another MI355X, tools/mfma_corner_probe.py): subnormal operands, underflowing terms,
    subnormal accumulators / results, signed zeros, Inf / NaN, overflow - the instruction on this box gives the fixture's bits, and the
    oracle (pinned on the fixture by tests/test_oracle_units.py) therefore restates this box too."""
    from helpers import load_golden
    from test_gpu_r5 import _hw_mfma
    g = load_golden('mfma_bf16_corners')
    d = _hw_mfma(g['a'], g['b'], g['c'])
    same = (d.view(np.uint32) == g['d_hw'].view(np.uint32)) | (np.isnan(d) & np.isnan(g['d_hw']))
    assert same.all(), {str(f): int((~same & (g['family'] == f)).sum()) for f in np.unique(g['family'][~same])}


def test_linear_bf16s_subnormal_split_terms_bit_exact(cuda_device):
    """The split GEMM on activations of magnitude 2^-126 .. 2^-100 (their second / third bf16 terms are subnormal or zero, many products
    and some results lie below the fp32 normal range): the kernel's own split (v_cvt_pk_bf16_f32 on the VALU) and the instruction against
    orc_linear_bf16s, bit for bit."""
    import gpu_utils as G
    from oracle import oracle as orc
    rng = np.random.default_rng(8)
    rows, cin, cout = 300, 128, 96
    x = (rng.standard_normal((rows, cin)) * np.exp2(rng.integers(-126, -100, (rows, 1)).astype(np.float64))).astype(np.float32)
    w = rng.standard_normal((cout, cin)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    yo = orc.linear_bf16s(x, w, b)
    y = G.conv2d(x.reshape(1, rows, 1, cin), np.ascontiguousarray(w.T).reshape(1, 1, cin, cout), b, 1, bf16s=True).reshape(rows, cout)
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), f'{(y.view(np.uint32) != yo.view(np.uint32)).sum()} of {y.size} differ'
    assert np.any((yo != 0) & (np.abs(yo) < 1.2e-38))


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
