"""-m gpu, round 4:
  * both sides of the Winograd-form size limit (the kernels address tensors with 32-bit byte offsets; a layer beyond the limit runs
    in the direct / phase-filter form - femasr_conv_wino_shape_ok, oracle.wino_fits): the limit is moved down to sizes a test can
    allocate with the per-handle femasr_debug_set_wino_limits hook, on the GPU and in the oracle together;
  * the whole reference testset/ directory through the CLI arithmetic (tests/golden/make_golden_r4.py), see test_testset_dir."""
import numpy as np
import pytest
import torch

from femasr_amd import _lib, synth
from helpers import oracle_net, synth_weights

pytestmark = pytest.mark.gpu


@pytest.fixture
def wino_limits():
    """(net, log2_total, log2_image) -> set on the net's native handle (femasr_debug_set_wino_limits: per handle since round 6) and on the
    oracle; the oracle's is restored afterwards, the handle dies with its net."""
    from oracle import oracle as orc
    prev = list(orc.WINO_LOG2_LIMITS)

    def set_(net, total, image):
        net.debug_wino_limits = (total, image)
        orc.WINO_LOG2_LIMITS[:] = [total or 31, image or 27]
    yield set_
    orc.WINO_LOG2_LIMITS[:] = prev


@pytest.mark.parametrize('log2_image,crossing', [(19, 'the 64-channel 96x96 layers and the x2 conv that makes them fall back'),
                                                 (18, 'the 128-channel 48x48 layers fall back too'),
                                                 (12, 'every layer falls back: the fp32_direct network')])
def test_winograd_size_limit_both_sides(cuda_device, wino_limits, log2_image, crossing):
    """x4 on a 32x32 input (padded to 48: decoder layers 256 @ 24^2 = 2^17.2, 128 @ 48^2 = 2^18.2, 64 @ 96^2 = 2^19.2 elements per
    image).  With the per-image limit between two of them, part of the decoder runs in the Winograd forms and part in the direct
    forms INSIDE one forward: the plan's GroupNorm-partial sizes, the weights each launch takes and the oracle's rule must agree
    layer by layer.  'fp32_strict': bit-identical to the oracle under the same limit; and within 1e-4 of the unlimited result."""
    import gpu_utils as G
    w = synth_weights('x4', 4, 'trained')
    x = synth.synth_input(77, (2, 3, 32, 32))
    xg = torch.from_numpy(x).to(cuda_device)
    net0 = G.build_net('x4', w, cuda_device, decoder_math='fp32_strict')
    net0.linear_math = 'fp32'         # (this test is about the conv forms: the fp32 chain keeps the CPU side of it 30x cheaper)
    y0, i0 = net0.test_with_indices(xg)
    y0, i0 = y0.cpu().numpy(), i0.cpu().numpy()
    del net0
    net = G.build_net('x4', w, cuda_device, decoder_math='fp32_strict')
    net.linear_math = 'fp32'
    wino_limits(net, 0, log2_image)
    y, idx = net.test_with_indices(xg)
    y, idx = y.cpu().numpy(), idx.cpu().numpy()
    yo, io = oracle_net('x4', w, linear_math='fp32').test(x, return_indices=True)
    assert np.array_equal(idx, io) and np.array_equal(idx, i0)
    assert np.array_equal(y, yo), f'limit 2^{log2_image} ({crossing}): max-abs vs oracle {np.abs(y - yo).max():.3e}'
    d = float(np.abs(y - y0).max())
    assert 0.0 < d < 1e-4, d          # a different algebraic form for some layers: different rounding, same function
    prof_names = [k for k in net.profile()] if hasattr(net, 'profile') else []
    print(f'limit 2^{log2_image}: {crossing}; max-abs vs the unlimited forward {d:.2e}; bit-identical to the oracle under the same limit', prof_names[:0])


def _testset():
    from helpers import load_golden
    return load_golden('testset_all')


def test_testset_dir(cuda_device, tmp_path):
    """BASELINE config 1 at full breadth: all 38 images of the reference's testset/ through the CLI arithmetic
    (inference_femasr.py:47-67: decode -> /255 -> test() or, from 600x600 pixels, test_tile(240, 16) -> clamp, x255, round), against
    what the REFERENCE produced for each (tests/golden/make_golden_r4.py; synthetic weights of seed 12, product default mode):
      * the VQ index map of every test() call under the near-tie rule (a different code only where the reference's own fp32
        distances have it within 4 ulp of its best; the receptive field of such a token is masked in the image checks),
      * the fp32 output (stored strided by 16) within 1e-3, its mean,
      * the uint8 image: equal up to 1 LSB on < 0.1 % of the pixels, and the SHA-256 of the whole image where no tie flipped.
    Then the CLI counterpart itself (femasr_amd.inference) over the directory: the PNG it writes for every PNG input must be the
    image this test just checked."""
    import io
    import hashlib
    from PIL import Image
    import gpu_utils as G
    from femasr_amd import imgproc, inference
    from helpers import weights_from_arch
    g = _testset()
    cfg = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
    seed, st = int(g['seed']), int(g['stride'])
    net = G.build_net('x4', weights_from_arch(cfg, seed, 'trained'), cuda_device, decoder_math='fp32')
    ts, pad = int(g['tile_size']), int(g['tile_pad'])
    near = {}
    for p_, cs, gs in zip(g['near_pos'], g['near_codes'], g['near_gaps_ulp']):
        near[int(p_)] = {int(c): float(gp) for c, gp in zip(cs, gs)}
    call = 0
    from oracle.near_tie import MAX_FLIPS_PER_IMAGE, MAX_FLIPS_TESTSET, NEAR_TIE_ULP, histogram
    flips_total, sha_equal, worst = 0, 0, 0.0
    flip_gaps = []
    expect_png = {}
    indir = tmp_path / 'in'
    indir.mkdir()
    for i, name in enumerate(g['names']):
        name = str(name)
        raw = g['file_bytes'][int(g['file_off'][i]):int(g['file_off'][i + 1])].tobytes()
        (indir / name).write_bytes(raw)
        rgb = np.asarray(Image.open(io.BytesIO(raw)).convert('RGB'))
        H, W = rgb.shape[:2]
        assert (H, W) == tuple(int(v) for v in g['hw'][i])
        x = imgproc.u8_to_input(torch.from_numpy(np.ascontiguousarray(rgb)).to(cuda_device))
        tiled = H * W >= 600 ** 2
        assert tiled == bool(g['tiled'][i])
        # the test() calls in the reference's order: the whole image, or its tiles row-major (femasr_arch.py:405-429)
        if tiled:
            y = net.test_tile(x, ts, pad)
            crops = []
            for ty in range(-(-H // ts)):
                for tx in range(-(-W // ts)):
                    y0, y1, x0, x1 = ty * ts, min(ty * ts + ts, H), tx * ts, min(tx * ts + ts, W)
                    y0p, x0p = max(y0 - pad, 0), max(x0 - pad, 0)
                    crops.append((y0p, x0p, x[:, :, y0p:min(y1 + pad, H), x0p:min(x1 + pad, W)].contiguous()))
            idxs = [(y0p, x0p, net.test_with_indices(c)[1]) for y0p, x0p, c in crops]
        else:
            y, idx = net.test_with_indices(x)
            idxs = [(0, 0, idx)]
        assert len(idxs) == int(g['n_calls'][i])
        assert tuple(y.shape) == tuple(int(v) for v in g['out_shape'][i])
        mask = np.zeros((4 * H, 4 * W), bool)
        off = int(g['token_off'][i])
        flips = 0
        for (y0p, x0p, idx) in idxs:
            hw = tuple(int(v) for v in g['call_hw'][call])
            call += 1
            assert tuple(idx.shape[-2:]) == hw
            got = idx.cpu().numpy().reshape(-1)
            ref = g['indices'][off:off + got.size].astype(np.int64)
            for r in np.nonzero(got != ref)[0]:
                gp = near.get(off + int(r), {}).get(int(got[r]), 1e9)
                assert gp <= NEAR_TIE_ULP, f'{name} token {r}: index {got[r]} vs reference {ref[r]}: {gp} ulp apart in the reference, not a near tie'
                flips += 1
                flip_gaps.append(gp)
                cy, cx = 4 * (y0p + 2 * (int(r) // hw[1])), 4 * (x0p + 2 * (int(r) % hw[1]))
                mask[max(cy - 128, 0):cy + 136, max(cx - 128, 0):cx + 136] = True
            off += got.size
        assert off == int(g['token_off'][i + 1])
        flips_total += flips
        yn = y.cpu().numpy()
        ms = mask[::st, ::st]
        f32 = g['f32_strided'][int(g['f32_off'][i]):int(g['f32_off'][i + 1])].reshape(3, *ms.shape)
        d = np.abs(yn[0, :, ::st, ::st] - f32).max(axis=0)
        if (~ms).any():
            err = float(d[~ms].max())
            worst = max(worst, err)
            assert err < 1e-3, (name, err, flips)
        out = imgproc.output_to_u8(y).cpu().numpy()
        u8 = g['u8_strided'][int(g['u8_off'][i]):int(g['u8_off'][i + 1])].reshape(*ms.shape, 3)
        du = np.abs(out[::st, ::st].astype(np.int16) - u8.astype(np.int16)).max(axis=2)
        if (~ms).any():
            assert du[~ms].max() <= 1 and (du[~ms] > 0).mean() < 2e-3, (name, int(du[~ms].max()), float((du[~ms] > 0).mean()))
        if flips == 0:
            assert abs(float(yn.astype(np.float64).mean()) - float(g['out_mean'][i])) < 1e-5, name
        sha_equal += int(hashlib.sha256(out.tobytes()).hexdigest() == str(g['out_sha256'][i]))
        if name.lower().endswith('.png'):
            expect_png[name] = out
    assert call == int(g['n_calls'].sum())
    assert flips_total <= MAX_FLIPS_TESTSET, flips_total          # (set from the measured count, see profiles/r05_parity_report.txt)
    # the CLI counterpart over the directory (PNG is lossless: what it wrote is what was checked above; a .jpg input is re-encoded)
    outdir = tmp_path / 'out'
    inference.main(['-i', str(indir), '-o', str(outdir), '-s', '4', '--synthetic-seed', str(seed)])
    for name in g['names']:
        name = str(name)
        im = np.asarray(Image.open(outdir / name).convert('RGB'))
        assert im.shape == (4 * int(g['hw'][list(g['names']).index(name)][0]), 4 * int(g['hw'][list(g['names']).index(name)][1]), 3)
        if name in expect_png:
            assert np.array_equal(im, expect_png[name]), name
    print(f'testset/: {len(g["names"])} images, {int(g["token_off"][-1])} tokens, {flips_total} accepted near-tie flips, max-abs fp32 vs the '
          f'reference outside their receptive fields {worst:.2e}, uint8 image bit-identical to the reference for {sha_equal} of {len(g["names"])}; '
          f'accepted gaps (ulp of the reference distance, rule <= {NEAR_TIE_ULP}): {histogram(flip_gaps)} [linear_math={net.linear_math}]')


def test_test_out_parameter_writes_in_place(cuda_device):
    """FeMaSRNet.test(x, out=buf): the forward's last kernel stores into the caller's tensor (all-gather send slabs, tile result
    buffers) - same bits as the allocating call, loud errors for a tensor that does not fit."""
    import gpu_utils as G
    net = G.build_net('x4', synth_weights('x4', 1, 'trained'), cuda_device)
    x = torch.from_numpy(synth.synth_input(9, (3, 3, 24, 40))).to(cuda_device)
    y = net.test(x)
    slab = torch.full((5, 3, 96, 160), -7.0, device=cuda_device)
    r = net.test(x, out=slab[1:4])
    assert r.data_ptr() == slab[1:4].data_ptr() and torch.equal(slab[1:4], y)
    assert float(slab[0].max()) == -7.0 and float(slab[4].max()) == -7.0          # nothing written outside the view
    for bad in (torch.empty((3, 3, 96, 161), device=cuda_device), torch.empty((3, 3, 96, 160), device=cuda_device, dtype=torch.float16),
                torch.empty((3, 3, 160, 96), device=cuda_device).transpose(2, 3)):
        with pytest.raises(ValueError):
            net.test(x, out=bad)
    # the tiled path produces its tiles in place and still equals the sequential reference loop
    img = torch.from_numpy(synth.synth_input(10, (1, 3, 70, 100))).to(cuda_device)
    net.max_tile_batch = 3
    yt = net.test_tile(img, 32, 8)
    from femasr_amd import tiling
    seq = torch.zeros_like(yt)
    for t in tiling.enumerate_tiles(70, 100, 32, 8):
        o = net.test(img[:, :, t.y0p:t.y1p, t.x0p:t.x1p].contiguous())
        ys, ye, xs, xe = t.out_src(4)
        a, b, c, d = t.out_dst(4)
        seq[:, :, a:b, c:d] = o[:, :, ys:ye, xs:xe]
    assert torch.equal(yt, seq)


@pytest.mark.parametrize('cin,cout,shape,nres', [(64, 128, (2, 9, 13), 0), (256, 128, (1, 12, 20), 1), (128, 64, (2, 16, 16), 0)])
def test_conv_up2_winograd_second_input(cuda_device, cin, cout, shape, nres):
    """femasr_conv_args.in_add: the x2 Winograd-type conv reads in + in_add (the decoder's `x = x + enc_feats[i]`, femasr_arch.py:361-362,
    added while the conv stages its input): bit-identical to the conv of the pre-added tensor; every other form refuses the field."""
    import gpu_utils as G
    from oracle import oracle as orc
    from femasr_amd._lib import FemasrError
    b, h, w = shape
    x = synth.uniform(41, 'ax', (b, h, w, cin), -2.0, 2.0)
    sk = synth.uniform(41, 'as', (b, h, w, cin), -1.0, 1.0)
    wt = synth.uniform(41, 'aw', (3, 3, cin, cout), -0.1, 0.1)
    bias = synth.uniform(41, 'ab', (cout,), -0.5, 0.5)
    r1 = synth.uniform(41, 'ar', (b, 2 * h, 2 * w, cout), -1, 1) if nres else None
    got, part = G.conv2d(x, wt, bias, 3, 1, 1, True, res1=r1, wino=True, gn_part=True, in_add=sk)
    ref = orc.conv2d((x + sk).astype(np.float32), wt, bias, 3, 1, 1, True, res1=r1, wino=True)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    same, part2 = G.conv2d((x + sk).astype(np.float32), wt, bias, 3, 1, 1, True, res1=r1, wino=True, gn_part=True)
    assert np.array_equal(got, same) and np.array_equal(part.cpu().numpy(), part2.cpu().numpy())
    with pytest.raises(FemasrError):
        G.conv2d(x, wt, bias, 3, 1, 1, True, in_add=sk)              # phase-filter form
    with pytest.raises(FemasrError):
        G.conv2d(x, wt, bias, 3, 1, 1, False, wino=True, in_add=sk)   # F(4x4,3x3) form
