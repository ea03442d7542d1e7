"""-m gpu, round 3: the HIP path against the full-size reference goldens of tests/golden/make_golden_r3.py -
  * the units of BASELINE configs 4 and 5 at their real sizes (x2 SR of a 256x256 tile; HQ autoencode of a 512x512 image), in the
    product default mode and in 'fp32_strict';
  * one image of the reference's testset/ that takes the CLI's `test_tile(240, 16)` branch (inference_femasr.py:58-63:
    OST_120.png, 720x720 -> 2880x2880, 9 tiles in 4 shape classes), uint8 in -> uint8 out.
Tolerance 1e-3 max-abs fp32 vs the reference; VQ indices exact (near-tie rule: zero mismatches today)."""
import hashlib
import io

import numpy as np
import pytest
import torch

from femasr_amd import synth
from helpers import cfg_name_of, check_indices_near_tie, load_golden, synth_weights, weights_from_arch

pytestmark = pytest.mark.gpu
TOL = 1e-3


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
    yn = y.cpu().numpy()
    err = float(np.abs(yn[:, :, ::8, ::8] - g['output_f32_stride8']).max())
    assert err < TOL, err
    assert abs(float(yn.astype(np.float64).mean()) - float(g['out_mean'])) < 1e-5
    out = imgproc.output_to_u8(y).cpu().numpy()
    diff = np.abs(out[::4, ::4].astype(np.int16) - g['output_u8_stride4'].astype(np.int16))
    # a value within ~1e-5 of x.5/255 may round the other way
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (int(diff.max()), float((diff > 0).mean()))
    same = hashlib.sha256(out.tobytes()).hexdigest() == str(g['output_u8_sha256'])
    print(f'OST_120 tiled [{math}]: max-abs vs reference {err:.3e}; uint8 image {"identical" if same else "differs in <= 1 LSB on %.2e of the strided pixels" % float((diff > 0).mean())}')
