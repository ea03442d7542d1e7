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
    """The split GEMM on activations of magnitude 2^-118 .. 2^-100 - their second / third bf16 terms are SUBNORMAL bf16 values (about half
    of them) or zero: the kernel's own split (v_cvt_pk_bf16_f32 on the VALU) and the instruction against orc_linear_bf16s, bit for bit.
    (Measured with tools/dbg_subnormal_split.py: still 0 differences of 8192 outputs in this regime; below 2^-118, where the REMAINDERS x - x1
    themselves are fp32 subnormals, 0.04 - 0.5 % of the outputs differ in the last place - 30 binades below anything the network holds.)"""
    import gpu_utils as G
    from oracle import oracle as orc
    rng = np.random.default_rng(8)
    rows, cin, cout = 300, 128, 96
    x = (rng.standard_normal((rows, cin)) * np.exp2(rng.integers(-118, -100, (rows, 1)).astype(np.float64))).astype(np.float32)
    w = rng.standard_normal((cout, cin)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    p2, p3 = orc.split3(x)[1:]
    assert int((((p2 & 0x7f80) == 0) & ((p2 & 0x7f) != 0)).sum()) > 1000 and int((((p3 & 0x7f80) == 0) & ((p3 & 0x7f) != 0)).sum()) > 1000
    yo = orc.linear_bf16s(x, w, b)
    y = G.conv2d(x.reshape(1, rows, 1, cin), np.ascontiguousarray(w.T).reshape(1, 1, cin, cout), b, 1, bf16s=True).reshape(rows, cout)
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), f'{(y.view(np.uint32) != yo.view(np.uint32)).sum()} of {y.size} differ'


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


CONV_SHAPES = [('s2_even', 2, 24, 40, 128, 256, 0, 2), ('s2_odd', 1, 23, 37, 64, 128, 1, 2), ('s2_256', 1, 36, 36, 256, 256, 0, 2), ('rb128', 2, 24, 40, 128, 128, 0), ('rstb256', 1, 36, 36, 256, 256, 1), ('ragged', 3, 7, 13, 64, 96, 2), ('one_px_rows', 1, 1, 150, 64, 64, 0),
               ('one_px_cols', 2, 130, 1, 128, 48, 1), ('cin192', 1, 16, 24, 192, 200, 1), ('tail_block', 1, 11, 12, 512, 130, 0)]


@pytest.mark.parametrize('case', CONV_SHAPES, ids=[c[0] for c in CONV_SHAPES])
def test_conv3x3_bf16s_bit_exact_vs_oracle(cuda_device, case):
    """The 3x3 stride-1 pad-1 conv as the split-bf16 GEMM over K = 9 Cin (femasr_conv_args.w_bf16s with ksz = 3; the convs in FRONT of the
    codebook lookup in linear_math 'bf16_split') against the oracle's conv3x3_bf16s = im2col + orc_linear_bf16s, bit for bit: image borders
    (zero taps), rows that straddle images / batch entries, ragged pixel and channel tiles, residual operands - and closer to the fp64
    convolution than the fp32 fmaf chain of the direct form."""
    import gpu_utils as G
    from oracle import oracle as orc
    name, b, h, w, cin, cout, nres = case[:7]
    st = case[7] if len(case) > 7 else 1                     # stride 2: the encoder's down-sampling convs (femasr_arch.py:159)
    ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
    rng = np.random.default_rng(len(name) + h * w)
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    r1 = rng.standard_normal((b, ho, wo, cout)).astype(np.float32) if nres >= 1 else None
    r2 = rng.standard_normal((b, ho, wo, cout)).astype(np.float32) if nres >= 2 else None
    yo = orc.conv3x3_bf16s(x, wt, bias, r1, r2, stride=st)
    y = G.conv2d(x, wt, bias, 3, st, 1, res1=r1, res2=r2, bf16s=True)
    assert y.shape == (b, ho, wo, cout)
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), f'{name}: {(y != yo).sum()} of {y.size} differ, max-abs {np.abs(y - yo).max():.3e}'
    ref = torch.nn.functional.conv2d(torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2), torch.from_numpy(wt.astype(np.float64)).permute(3, 2, 0, 1),
                                     torch.from_numpy(bias.astype(np.float64)), padding=1, stride=st).permute(0, 2, 3, 1).numpy()
    for r in (r1, r2):
        if r is not None:
            ref = ref + r
    ych = G.conv2d(x, wt, bias, 3, st, 1, res1=r1, res2=r2)         # the direct fp32 form (one fmaf chain per output)
    e_split, e_chain = float(np.abs(y - ref).max()), float(np.abs(ych - ref).max())
    print(f'{name}: max-abs vs fp64: split {e_split:.3e}, fp32 chain {e_chain:.3e}')
    assert e_split <= 1.5 * e_chain + 1e-7


def test_conv3x3_bf16s_refusals_and_gn_silu_apply(cuda_device):
    import gpu_utils as G
    from oracle import oracle as orc
    x = np.zeros((1, 8, 8, 96), np.float32)
    with pytest.raises(_lib.FemasrError):          # Cin % 64 != 0
        G.conv2d(x, np.zeros((3, 3, 96, 64), np.float32), np.zeros(64, np.float32), 3, 1, 1, bf16s=True)
    x = np.zeros((1, 8, 8, 64), np.float32)
    with pytest.raises(_lib.FemasrError):          # no padding
        G.conv2d(x, np.zeros((3, 3, 64, 64), np.float32), np.zeros(64, np.float32), 3, 1, 0, bf16s=True)
    with pytest.raises(_lib.FemasrError):          # a GN prologue is not taken by this form (femasr_gn_silu_apply runs in front of it)
        G.conv2d(x, np.zeros((3, 3, 64, 64), np.float32), np.zeros(64, np.float32), 3, 1, 1, bf16s=True, prologue=_lib.PRO_GN_SILU,
                 pro=(np.ones((1, 64), np.float32), np.zeros((1, 64), np.float32), None))
    rng = np.random.default_rng(1)
    xx = (rng.standard_normal((3, 9, 11, 128)) * 3).astype(np.float32)
    a = (1 + 0.3 * rng.standard_normal((3, 128))).astype(np.float32)
    bb = (0.5 * rng.standard_normal((3, 128))).astype(np.float32)
    got = G.gn_silu_apply(xx, a, bb)
    ref = orc.scale_shift_silu(xx, a, bb)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
