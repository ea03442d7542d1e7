"""-m gpu, round 3: the HIP path against the full-size reference goldens of tests/golden/make_golden_r3.py -
  * the units of BASELINE configs 4 and 5 at their real sizes (x2 SR of a 256x256 tile; HQ autoencode of a 512x512 image), in the
    product default mode and in 'fp32_strict';
  * one image of the reference's testset/ that takes the CLI's `test_tile(240, 16)` branch (inference_femasr.py:58-63:
    OST_120.png, 720x720 -> 2880x2880, 9 tiles in 4 shape classes), uint8 in -> uint8 out.
Tolerance 1e-3 max-abs fp32 vs the reference; VQ indices exact up to the documented near-tie rule (zero mismatches on the
single-tile cases; the 173 056-token image holds reference near ties, see the test)."""
import hashlib
import io

import numpy as np
import pytest
import torch

from femasr_amd import synth
from helpers import cfg_name_of, check_indices_near_tie, load_golden, synth_weights, weights_from_arch

pytestmark = pytest.mark.gpu
TOL = 1e-3
# A VQ index other than the reference's is accepted only where the REFERENCE's own fp32 distances of the two codes are this close
# (d ~ 500 there: 4 ulp = 1.2e-4 absolute, 2.4e-7 relative - the size of the encoders' summation-order differences; SURVEY 7, hard
# part 1).  The single-tile goldens need no allowance at all; the 173 056-token image has 193 tokens within 16 ulp of a tie.
# Measured (rounds 3 and 4): 2 such tokens, gaps 0 and 3 ulp - the bar sits at what is measured (was 8 ulp / 40 flips).
from oracle.near_tie import NEAR_TIE_ULP      # the one rule (oracle/near_tie.py)
MAX_FLIPS = 4


@pytest.mark.parametrize('name', ['x2_tile256_trained', 'hq_full512_trained'])
@pytest.mark.parametrize('math', ['fp32', 'fp32_strict'])
def test_full_size_units_vs_reference(cuda_device, name, math):
    import gpu_utils as G
    g = load_golden(name)
    cn = cfg_name_of(g)
    net = G.build_net(cn, synth_weights(cn, int(g['seed']), str(g['codebook'])), cuda_device, decoder_math=math)
    x = torch.from_numpy(synth.synth_input(int(g['input_seed']), tuple(g['in_shape']))).to(cuda_device)
    if str(g['mode']) == 'test':
        y, idx = net.test_with_indices(x)
    else:
        y, _, _, il = net(x)
        idx = il[0]
    y, idx = y.cpu().numpy(), idx.cpu().numpy()
    st = int(g['out_stride'])
    assert y.shape == tuple(g['out_shape'])
    err = float(np.abs(y[:, :, ::st, ::st] - g['output']).max())
    assert err < TOL, err
    assert abs(float(y.astype(np.float64).mean()) - float(g['out_mean'])) < 1e-5
    nbad, _ = check_indices_near_tie(idx, g)
    assert nbad == 0
    print(f'{name} [{math}]: max-abs vs reference {err:.3e} (output absmax {float(g["out_absmax"]):.3g}), indices exact')


@pytest.mark.parametrize('math', ['fp32', 'fp32_strict'])
def test_cli_tiled_branch_on_testset_png(cuda_device, math):
    """inference_femasr.py:50-67 on OST_120.png: h*w >= 600^2 -> net.test_tile(img) with the defaults (240, 16), then tensor2img."""
    from PIL import Image
    import gpu_utils as G
    from femasr_amd import imgproc
    g = load_golden('png_OST_120_tiled')
    cfg = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
    w = weights_from_arch(cfg, int(g['seed']), 'trained')
    net = G.build_net('x4', w, cuda_device, decoder_math=math)
    rgb = np.asarray(Image.open(io.BytesIO(g['png'].tobytes())).convert('RGB'))
    assert rgb.shape[0] * rgb.shape[1] >= 600 ** 2
    u8 = torch.from_numpy(np.ascontiguousarray(rgb)).to(cuda_device)
    x = imgproc.u8_to_input(u8)                            # (1,3,H,W) fp32 in [0,1], on the GPU
    y = net.test_tile(x, int(g['tile_size']), int(g['tile_pad']))
    assert tuple(y.shape) == tuple(g['out_shape'])
    # The VQ index map of every tile, in the reference's loop order (femasr_arch.py:405-429), under the documented near-tie rule
    # (helpers.check_indices_near_tie): 173 056 tokens with synthetic weights hold a few dozen whose two best codes are within
    # 2 ulp of each other in the REFERENCE's own distances (sometimes more than two codes: large |z|^2 quantises d); such a token may
    # resolve to any of the codes the reference itself has within 2 ulp of its best.  Anything else fails.
    ts, pad = int(g['tile_size']), int(g['tile_pad'])
    H, W = x.shape[2:]
    near = {int(p): {int(c): float(gp) for c, gp in zip(cs, gs)} for p, cs, gs in zip(g['near_tie_pos'], g['near_tie_codes'], g['near_tie_gaps_ulp'])}
    flip_gaps = []
    mask = np.zeros((4 * H, 4 * W), bool)          # output pixels that may differ because an accepted near tie went the other way
    R = 128                                          # decoder receptive radius in output pixels, generous
    off = k = flips = 0
    for ty in range(-(-H // ts)):
        for tx in range(-(-W // ts)):
            y0, y1, x0, x1 = ty * ts, min(ty * ts + ts, H), tx * ts, min(tx * ts + ts, W)
            y0p, x0p = max(y0 - pad, 0), max(x0 - pad, 0)
            crop = x[:, :, y0p:min(y1 + pad, H), x0p:min(x1 + pad, W)].contiguous()
            _, idx = net.test_with_indices(crop)
            hw = tuple(int(v) for v in g['tile_index_hw'][k])
            assert tuple(idx.shape[-2:]) == hw
            n = hw[0] * hw[1]
            got = idx.cpu().numpy().reshape(-1)
            ref = g['tile_indices'][off:off + n].astype(np.int64)
            for r in np.nonzero(got != ref)[0]:
                gp = near.get(off + int(r), {}).get(int(got[r]), 1e9)
                assert gp <= NEAR_TIE_ULP, f'tile ({ty},{tx}) token {r}: index {got[r]} vs reference {ref[r]}: {gp} ulp apart in the reference, not a near tie'
                flip_gaps.append(gp)
                flips += 1
                cy, cx = 4 * (y0p + 2 * (int(r) // hw[1])), 4 * (x0p + 2 * (int(r) % hw[1]))      # a token = 2x2 LR pixels = 8x8 output pixels
                mask[max(cy - R, 0):cy + 8 + R, max(cx - R, 0):cx + 8 + R] = True
            off += n
            k += 1
    assert off == g['tile_indices'].size
    # (the mask bound follows from the flip allowance: MAX_FLIPS receptive fields of (2 R + 8)^2 output pixels; it was a literal 0.02 while
    # the image had 2 flips - round 6, with the 3x3 convs in front of the lookup in the split arithmetic, resolves 3 reference ties differently)
    mask_bound = MAX_FLIPS * (2 * R + 8) ** 2 / float(16 * H * W) + 1e-3
    print(f'OST_120 tiled [{math}]: {flips} flips at reference gaps {sorted(flip_gaps)} ulp, {100 * mask.mean():.2f} % of the output masked (bound {100 * mask_bound:.2f} %)')
    assert flips <= MAX_FLIPS and mask.mean() < mask_bound, (flips, float(mask.mean()))
    yn = y.cpu().numpy()
    dz = np.abs(yn[:, :, ::8, ::8] - g['output_f32_stride8']).max(axis=(0, 1))
    err = float(dz[~mask[::8, ::8]].max())
    assert err < TOL, (err, flips)
    if flips == 0:
        assert abs(float(yn.astype(np.float64).mean()) - float(g['out_mean'])) < 1e-5
    out_t = imgproc.output_to_u8(y)
    # round 6: the uint8 tile path (crops, forwards and paste on bytes; what the CLI's tiled branch calls) gives the same image, bit for bit
    out_u8 = net.test_tile_u8(u8, ts, pad)
    assert out_u8.dtype == torch.uint8 and torch.equal(out_u8, out_t), int((out_u8 != out_t).sum())
    out = out_t.cpu().numpy()
    diff = np.abs(out[::4, ::4].astype(np.int16) - g['output_u8_stride4'].astype(np.int16)).max(axis=2)[~mask[::4, ::4]]
    # a value within ~1e-5 of x.5/255 may round the other way
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (int(diff.max()), float((diff > 0).mean()))
    same = hashlib.sha256(out.tobytes()).hexdigest() == str(g['output_u8_sha256'])
    print(f'OST_120 tiled [{math}]: {flips} accepted near-tie flips, reference gaps {sorted(flip_gaps)} ulp ({100 * mask.mean():.2f} % of the output masked); max-abs vs reference outside {err:.3e}; '
          f'uint8 image {"identical" if same else "differs in <= 1 LSB on %.2e of the unmasked strided pixels" % float((diff > 0).mean())}')
