"""-m gpu: the whole hot path (FeMaSRNet on the HIP library) against
  (1) the CPU oracle on the same seeded weights/inputs — BIT-EXACT outputs and VQ indices in decoder_math='fp32_strict'
      (what gpu_utils.build_net selects) and 'fp32_direct'; the product default 'fp32' (SiLU of the Winograd convs on the
      hardware exp2 / rcp units) has bit-exact VQ indices and an image within 1e-4 of the oracle's, and
  (2) the committed golden vectors recorded from the reference — outputs within 1e-3 max-abs fp32,
      indices exact (near-tie rule of helpers.check_indices_near_tie; currently zero mismatches),
plus size-independent properties at BASELINE.json's full sizes (batch invariance, determinism,
tiled == batched, decode(encode) consistency)."""
import numpy as np
import pytest
import torch

from femasr_amd import synth
from helpers import cfg_name_of, check_indices_near_tie, load_golden, oracle_net, synth_weights

pytestmark = pytest.mark.gpu
TOL = 1e-3       # north-star tolerance vs the reference forward (max-abs, fp32)


def _case(name):
    import gpu_utils as G
    g = load_golden(name)
    cn = cfg_name_of(g)
    w = synth_weights(cn, int(g['seed']), str(g['codebook']))
    x = synth.synth_input(int(g['input_seed']), tuple(g['in_shape']))
    return g, cn, w, x, G.build_net(cn, w)


@pytest.mark.parametrize('name', ['x4_small_init', 'x4_small_trained', 'x2_small_trained'])
def test_test_path_vs_oracle_and_reference(cuda_device, name):
    g, cn, w, x, net = _case(name)
    y, idx = net.test_with_indices(torch.from_numpy(x).cuda())
    y, idx = y.cpu().numpy(), idx.cpu().numpy()
    assert y.shape == tuple(g['out_shape']) and idx.dtype == np.int64 and idx.shape == g['vq_indices'].shape
    # (2) reference goldens
    assert np.abs(y - g['output']).max() < TOL
    nbad, nacc = check_indices_near_tie(idx, g)
    assert nbad == 0, f'{nbad} VQ index mismatches vs the reference ({nacc} near-tie acceptable)'
    # (1) oracle, bit-exact
    yo, io = oracle_net(cn, w).test(x, return_indices=True)
    assert np.array_equal(idx, io), f'{np.sum(idx != io)} index mismatches vs oracle'
    assert np.array_equal(y, yo), f'not bit-identical to the oracle: max-abs {np.abs(y - yo).max():.3e}'


@pytest.mark.parametrize('name', ['x4_small_init', 'x4_small_trained', 'x2_small_trained', 'hq_small_trained', 'x4_tile128_trained'])
def test_default_mode_vs_oracle_and_reference(cuda_device, name):
    """The product default decoder_math='fp32': identical VQ indices (everything that feeds a codebook lookup is exact), the
    image within 1e-4 max-abs of the oracle's and within the north-star 1e-3 of the reference golden - and really a different
    kernel path from fp32_strict."""
    g, cn, w, x, net = _case(name)
    xt = torch.from_numpy(x).cuda()
    run = (lambda: net.test_with_indices(xt)) if cn != 'hq' else (lambda: (lambda o: (o[0], o[3][0]))(net(xt)))
    ys, ids = run()
    net.decoder_math = 'fp32'
    yd, idd = run()
    assert torch.equal(ids, idd)
    dev = float((yd - ys).abs().max())
    assert 0.0 < dev < 1e-4, dev
    st = int(g['out_stride']) if 'out_stride' in g else 1
    yd = yd.cpu().numpy()
    assert np.abs(yd[:, :, ::st, ::st] - g['output']).max() < TOL
    nbad, _ = check_indices_near_tie(idd.cpu().numpy(), g)
    assert nbad == 0


def test_hq_forward_and_decode_indices(cuda_device):
    g, cn, w, x, net = _case('hq_small_trained')
    out, cl, sl, idx_list = net(torch.from_numpy(x).cuda())
    assert float(cl) == 0.0 and float(sl) == 0.0 and len(idx_list) == 1
    y, idx = out.cpu().numpy(), idx_list[0].cpu().numpy()
    assert np.abs(y - g['output']).max() < TOL
    assert np.array_equal(idx, g['vq_indices'])
    yo, io = oracle_net(cn, w).forward(x)
    assert np.array_equal(idx, io) and np.array_equal(y, yo)
    # decode_indices(indices): bit-exact vs the oracle's decode_indices; vs forward() only CLOSE, because forward feeds
    # the straight-through z + (e - z) (femasr_arch.py:95), which is not bit-equal to e (hard part 2)
    y2 = net.decode_indices(idx_list[0]).cpu().numpy()
    assert np.array_equal(y2, oracle_net(cn, w).decode_indices(idx))
    assert np.abs(y2 - y).max() < 1e-5
    # and matches the reference's decode_indices golden
    gd = load_golden('hq_decode_indices')
    import gpu_utils as G
    net2 = G.build_net('hq', synth_weights('hq', int(gd['seed']), str(gd['codebook'])))
    yd = net2.decode_indices(torch.from_numpy(gd['indices'])).cpu().numpy()
    assert np.abs(yd - gd['output']).max() < TOL
    with pytest.raises(AssertionError):
        net2.decode_indices(torch.zeros(4, 4, dtype=torch.int64))


def test_test_tile_vs_reference_and_oracle(cuda_device):
    g, cn, w, x, net = _case('x4_tiled_trained')
    ts, pad = int(g['kw_tile_size']), int(g['kw_tile_pad'])
    y = net.test_tile(torch.from_numpy(x).cuda(), ts, pad).cpu().numpy()
    assert y.shape == tuple(g['out_shape'])
    assert np.abs(y - g['output']).max() < TOL
    yo = oracle_net(cn, w).test_tile(x, ts, pad)
    assert np.array_equal(y, yo), f'max-abs vs oracle {np.abs(y - yo).max():.3e}'
    # two "ranks" in one process: partition + gather is a pure function of (tiles, world)
    from femasr_amd import tiling
    xt = torch.from_numpy(x).cuda()

    def fake_gather(results, classes, batch, channel, scale):
        other = {}
        for hw, tl in tiling.partition(classes, 1, 2).items():
            other[hw] = torch.cat([net.test(xt[:, :, t.y0p:t.y1p, t.x0p:t.x1p]) for t in tl], 0) if tl else \
                xt.new_zeros((0, channel, hw[0] * scale, hw[1] * scale))
        return [results, other]
    y2 = net.test_tile(xt, ts, pad, rank=0, world_size=2, gather=fake_gather).cpu().numpy()
    assert np.array_equal(y2, y)


def test_full_size_tile_vs_reference(cuda_device):
    """BASELINE config-2 geometry: 128x128 LR -> (144 padded) -> 512x512, both codebook regimes."""
    for name in ('x4_tile128_init', 'x4_tile128_trained'):
        g, cn, w, x, net = _case(name)
        xb = torch.from_numpy(np.concatenate([x, x[:, :, ::-1].copy()], 0)).cuda()      # B=2: golden + a flipped copy
        y, idx = net.test_with_indices(xb)
        y, idx = y.cpu().numpy(), idx.cpu().numpy()
        st = int(g['out_stride'])
        assert np.abs(y[:1, :, ::st, ::st] - g['output']).max() < TOL
        assert abs(float(y[:1].astype(np.float64).mean()) - float(g['out_mean'])) < 1e-5
        nbad, nacc = check_indices_near_tie(idx[:1], g)
        assert nbad == 0, f'{name}: {nbad} index mismatches vs the reference ({nacc} near-tie acceptable)'
        del net
        torch.cuda.empty_cache()


def test_full_size_batch16_properties(cuda_device):
    """x4, batch 16 of 128x128 tiles (the benchmarked workload): size-independent properties, in the mode that is bit-identical to
    the oracle and in the product default mode that bench.py times; tiles 0 / 15 also against the CPU oracle (strict: same
    bits; default: indices exact, image within 1e-4).  One test for both modes: the two oracle forwards (~20 s of CPU each) are
    computed once (helpers.oracle_net memoises them)."""
    for math in ('fp32_strict', 'fp32'):
        _batch16_properties(math)


def _batch16_properties(math):
    import gpu_utils as G
    w = synth_weights('x4', 0, 'trained')
    net = G.build_net('x4', w, decoder_math=math)
    x = torch.from_numpy(synth.synth_input(5, (16, 3, 128, 128))).cuda()
    y, idx = net.test_with_indices(x)
    assert y.shape == (16, 3, 512, 512) and idx.shape == (16, 1, 72, 72) and torch.isfinite(y).all()
    # determinism: same bits twice
    y2, idx2 = net.test_with_indices(x)
    assert torch.equal(y, y2) and torch.equal(idx, idx2)
    # batch invariance: sample i alone == sample i inside the batch (all ops are per-sample)
    onet = oracle_net('x4', w)
    for i in (0, 7, 15):
        yi, ii = net.test_with_indices(x[i:i + 1])
        assert torch.equal(yi[0], y[i]) and torch.equal(ii[0], idx[i])
        if i == 7:
            continue                  # (batch invariance only: two oracle tiles are enough)
        yo, io = onet.test(x[i:i + 1].cpu().numpy(), return_indices=True)
        assert np.array_equal(ii.cpu().numpy(), io), f'tile {i}: VQ indices differ from the oracle'
        err = float(np.abs(yi.cpu().numpy() - yo).max())
        assert err <= (0.0 if math == 'fp32_strict' else 1e-4), (math, i, err)
    # tiled == batched: a 512x512 image cut into 16 tiles of 128 with no halo IS the batch of its crops
    img = torch.cat([torch.cat([x[r * 4 + c] for c in range(4)], 2) for r in range(4)], 1)[None]
    yt = net.test_tile(img, 128, 0)
    for r in range(4):
        for c in range(4):
            assert torch.equal(yt[0, :, r * 512:(r + 1) * 512, c * 512:(c + 1) * 512], y[r * 4 + c])
    # sub-batch streams (femasr_set_streams): same bits for any split, incl. an uneven one (16 = 6+5+5)
    for ns in (2, 3):
        net.num_streams = ns
        ys, idxs = net.test_with_indices(x)
        assert torch.equal(ys, y) and torch.equal(idxs, idx), ns
    net.num_streams = 1
    # indices index real codes; decode path accepts them
    assert int(idx.min()) >= 0 and int(idx.max()) < 1024


def test_weight_updates_are_picked_up_and_errors_are_loud(cuda_device):
    import gpu_utils as G
    from femasr_amd._lib import FemasrError
    w = synth_weights('hq', 3, 'trained')
    net = G.build_net('hq', w)
    x = torch.from_numpy(synth.synth_input(9, (1, 3, 32, 32))).cuda()
    y0 = net(x)[0].clone()
    with torch.no_grad():
        net.out_conv.bias.add_(1.0)
    y1 = net(x)[0]
    assert torch.allclose(y1, y0 + 1.0, atol=1e-5) and not torch.equal(y1, y0)
    with pytest.raises(ValueError):
        net.test(torch.zeros(1, 4, 32, 32, device='cuda'))
    with pytest.raises(FemasrError):
        net.test(torch.zeros(1, 3, 32, 32))            # CPU tensor: no fallback
    # HQ stage (no Swin): like the reference, forward() takes any size (30 -> 29 -> 15 -> 8 -> 4 -> 32 rows out) ...
    w = dict(w)
    w['out_conv.bias'] = w['out_conv.bias'] + np.float32(1.0)          # the in-place update made above
    x30 = synth.synth_input(10, (1, 3, 30, 32))
    y30 = net(torch.from_numpy(x30).cuda())[0].cpu().numpy()
    yo, _ = oracle_net('hq', w).forward(x30)
    assert y30.shape == yo.shape == (1, 3, 32, 32) and np.array_equal(y30, yo)
    # ... and test() on an image smaller than its mirror pad runs on the truncated flip-concat (femasr_arch.py:459-460)
    x17 = synth.synth_input(11, (1, 3, 17, 20))
    y17 = net.test(torch.from_numpy(x17).cuda()).cpu().numpy()
    assert np.array_equal(y17, oracle_net('hq', w).test(x17))
    # the LQ stage keeps the Swin divisibility requirement (the reference fails in window_partition)
    lq = G.build_net('x4', synth_weights('x4', 3, 'trained'))
    with pytest.raises(FemasrError):
        lq(torch.zeros(1, 3, 30, 32, device='cuda'))


def test_image_pre_post_kernels_exact(cuda_device):
    from femasr_amd import imgproc
    from oracle import oracle as orc
    u8 = (synth.uniform01(1, 'img', 37 * 53 * 3) * 256).astype(np.uint8).reshape(37, 53, 3)
    for bgr in (False, True):
        x = imgproc.u8_to_input(torch.from_numpy(u8).cuda(), bgr=bgr).cpu().numpy()
        assert np.array_equal(x, orc.image_u8_to_f32(u8, bgr=bgr))
        y = synth.uniform(2, 'out', (1, 3, 41, 29), -0.3, 1.3)
        y[0, 0, 0, :4] = np.array([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255], np.float32)
        got = imgproc.output_to_u8(torch.from_numpy(y).cuda(), bgr=bgr).cpu().numpy()
        assert np.array_equal(got, orc.image_f32_to_u8(y, bgr=bgr))


def test_cli_counterpart_end_to_end(cuda_device, tmp_path):
    """femasr_amd/inference.py == inference_femasr.py:19-69 policy: sorted files, h*w < max_size^2 -> test(), else
    test_tile(); uint8 PNG out.  Expected pixels = oracle pre -> oracle net -> oracle post (bit-exact chain)."""
    from PIL import Image
    from femasr_amd import inference
    from oracle import oracle as orc
    src, dst = tmp_path / 'in', tmp_path / 'out'
    src.mkdir()
    imgs = {}
    for name, (h, w) in (('b_small.png', (24, 40)), ('a_large.png', (48, 40))):
        u8 = (synth.synth_input(11, (1, 3, h, w), tag=name)[0].transpose(1, 2, 0) * 255).astype(np.uint8)
        Image.fromarray(u8, 'RGB').save(src / name)
        imgs[name] = u8
    dst_d = tmp_path / 'out_default'
    base = ['-i', str(src), '-s', '4', '--synthetic-seed', '1', '--max_size', '40', '--tile_size', '24', '--tile_pad', '8']
    inference.main(base + ['-o', str(dst), '--decoder-math', 'fp32_strict'])
    inference.main(base + ['-o', str(dst_d)])                  # the product default ('fp32': hardware-transcendental SiLU behind the lookup)
    w = synth_weights('x4', 1, 'trained')
    onet = oracle_net('x4', w)
    for name, u8 in imgs.items():
        x = orc.image_u8_to_f32(u8)
        h, wd = u8.shape[:2]
        y = onet.test(x) if h * wd < 40 ** 2 else onet.test_tile(x, 24, 8)
        expect = orc.image_f32_to_u8(y)
        got = np.asarray(Image.open(dst / name).convert('RGB'))
        assert got.shape == (h * 4, wd * 4, 3)
        assert np.array_equal(got, expect), name
        got_d = np.asarray(Image.open(dst_d / name).convert('RGB'))
        diff = np.abs(got_d.astype(np.int16) - expect.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (name, int(diff.max()), float((diff > 0).mean()))


def test_other_configs_full_size_properties(cuda_device):
    """BASELINE configs 4 and 5 at (reduced-batch) full geometry: x2 256x256 -> 512x512, HQ autoencode 512x512."""
    import gpu_utils as G
    net2 = G.build_net('x2', synth_weights('x2', 2, 'trained'))
    x = torch.from_numpy(synth.synth_input(6, (4, 3, 256, 256))).cuda()
    y, idx = net2.test_with_indices(x)
    assert y.shape == (4, 3, 512, 512) and idx.shape == (4, 1, 72, 72) and torch.isfinite(y).all()
    y1, i1 = net2.test_with_indices(x[2:3])
    assert torch.equal(y1[0], y[2]) and torch.equal(i1[0], idx[2])
    del net2
    torch.cuda.empty_cache()
    neth = G.build_net('hq', synth_weights('hq', 3, 'trained'))
    xh = torch.from_numpy(synth.synth_input(7, (2, 3, 512, 512))).cuda()
    out, _, _, il = neth(xh)
    assert out.shape == (2, 3, 512, 512) and il[0].shape == (2, 1, 64, 64) and torch.isfinite(out).all()
    o1 = neth(xh[1:2])[0]
    assert torch.equal(o1[0], out[1])
    assert (neth.decode_indices(il[0]) - out).abs().max() < 1e-5      # straight-through z+(e-z) vs e: rounding-level only


def test_decoder_bf16x3_mode(cuda_device):
    """Opt-in decoder_math='bf16x3': the convs behind the codebook lookup run on the bf16 matrix cores (3-term split).
    VQ indices must be IDENTICAL to the exact mode (everything feeding the argmin stays fp32); outputs must stay
    within the north-star bound 1e-3 max-abs of the reference golden and of the oracle."""
    for name in ('x4_small_trained', 'x4_tile128_trained'):
        g, cn, w, x, net = _case(name)
        xt = torch.from_numpy(x).cuda()
        y32, i32 = net.test_with_indices(xt)
        net.decoder_math = 'bf16x3'
        y16, i16 = net.test_with_indices(xt)
        assert torch.equal(i16, i32)
        assert not torch.equal(y16, y32)                 # the other kernels really ran
        err_exact = float((y16 - y32).abs().max())
        st = int(g['out_stride'])
        err_ref = float(np.abs(y16.cpu().numpy()[:, :, ::st, ::st] - g['output']).max())
        print(f'{name}: bf16x3 vs fp32 path {err_exact:.3e}, vs reference golden {err_ref:.3e}')
        assert err_exact < TOL and err_ref < TOL
        nbad, _ = check_indices_near_tie(i16.cpu().numpy(), g)
        assert nbad == 0
        net.decoder_math = 'fp32_strict'
        y32b, _ = net.test_with_indices(xt)
        assert torch.equal(y32b, y32)
        del net
        torch.cuda.empty_cache()


@pytest.mark.parametrize('name', ['x4_small_trained', 'x2_small_trained'])
def test_fp32_direct_mode_vs_oracle(cuda_device, name):
    """decoder_math='fp32_direct': every conv in the direct form (no Winograd behind the codebook lookup) -- bit-identical to
    OracleNet(winograd=False), same VQ indices as the default mode, output within fp32 rounding of it, and the default mode
    is restored bit-for-bit afterwards."""
    from oracle import oracle as orc
    from helpers import CONFIGS
    g, cn, w, x, net = _case(name)
    xt = torch.from_numpy(x).cuda()
    y0, i0 = net.test_with_indices(xt)
    net.decoder_math = 'fp32_direct'
    y1, i1 = net.test_with_indices(xt)
    assert torch.equal(i0, i1)
    assert not torch.equal(y0, y1)                       # the other kernels really ran
    assert float((y0 - y1).abs().max()) < 1e-4
    cfg = CONFIGS[cn]
    ref = orc.OracleNet(w, codebook_params=cfg['codebook_params'], LQ_stage=cfg['LQ_stage'], scale_factor=cfg.get('scale_factor', 4),
                        winograd=False)
    yo, io = ref.test(x, return_indices=True)
    assert np.array_equal(i1.cpu().numpy(), io)
    assert np.array_equal(y1.cpu().numpy(), yo), f'max-abs {np.abs(y1.cpu().numpy() - yo).max():.3e}'
    assert np.abs(y1.cpu().numpy() - g['output']).max() < TOL
    net.decoder_math = 'fp32_strict'
    y2, _ = net.test_with_indices(xt)
    assert torch.equal(y2, y0)
