"""Host-side (no GPU) tests of the inference-only FeMaSRModel surface, the folder datasets and the option parser
(YAML -> build_network -> validation loop; checkpoint format round trip)."""
import os

import numpy as np
import pytest
import torch
import yaml

from femasr_amd import data as fdata
from femasr_amd.archs import build_network
from femasr_amd.models import MODEL_REGISTRY, build_model
from femasr_amd.models import femasr_model as fm
from femasr_amd.test import parse_options
from helpers import CONFIGS


def _png(path, arr):
    from PIL import Image
    Image.fromarray(arr, 'RGB').save(path)


def test_model_registry_and_guards():
    assert 'FeMaSRModel' in MODEL_REGISTRY
    opt = dict(model_type='FeMaSRModel', name='t', is_train=True, network_g=dict(type='FeMaSRNet', **CONFIGS['x4']))
    with pytest.raises(NotImplementedError):
        build_model(opt)
    if not torch.cuda.is_available():       # the product path has no CPU fallback
        opt['is_train'] = False
        with pytest.raises(RuntimeError, match='GPU'):
            build_model(opt)


def test_load_network_reference_semantics(tmp_path):
    """base_model.py:291-323: 'params' key, 'module.' prefix stripped, non-strict load skips size mismatches."""
    net = build_network(dict(type='FeMaSRNet', **CONFIGS['x4']))
    sd = {k: torch.full_like(v, 0.5) if v.is_floating_point() else v.clone() for k, v in net.state_dict().items()}
    bad_key = 'out_conv.bias'
    sd[bad_key] = torch.zeros(7)                                    # wrong size
    ckpt = {'params': {'module.' + k: v for k, v in sd.items()}}
    path = str(tmp_path / 'net_g.pth')
    torch.save(ckpt, path)
    before = net.state_dict()[bad_key].clone()
    res = fm.FeMaSRModel.load_network(None, net, path, strict=False, param_key='params_ema')   # falls back to 'params'
    assert bad_key in res.missing_keys and (bad_key + '.ignore') in res.unexpected_keys
    after = net.state_dict()
    assert torch.equal(after[bad_key], before)
    k0 = 'multiscale_encoder.in_conv.weight'
    assert float(after[k0].flatten()[0]) == 0.5
    with pytest.raises(RuntimeError):                                # strict: the size mismatch is an error
        fm.FeMaSRModel.load_network(None, net, path, strict=True)
    with pytest.raises(RuntimeError, match='no network'):
        fm.FeMaSRModel.load_network(None, net, 'https://example.invalid/x.pth')


def test_folder_datasets_and_options(tmp_path):
    rng = np.random.RandomState(0)
    lq, gt = tmp_path / 'lq', tmp_path / 'gt'
    lq.mkdir(); gt.mkdir()
    for name in ('b.png', 'a.png', 'c.png'):
        _png(str(lq / name), rng.randint(0, 256, (8, 12, 3), dtype=np.uint8))
        _png(str(gt / name), rng.randint(0, 256, (32, 48, 3), dtype=np.uint8))
    (lq / 'notes.txt').write_text('not an image')
    ds = fdata.build_dataset(dict(type='PairedImageDataset', name='d', dataroot_lq=str(lq), dataroot_gt=str(gt)))
    assert len(ds) == 3 and [os.path.basename(p) for p in ds.lq_paths] == ['a.png', 'b.png', 'c.png']
    item = ds[1]
    assert item['lq'].shape == (3, 8, 12) and item['gt'].shape == (3, 32, 48) and item['lq'].dtype == torch.float32
    assert 0.0 <= float(item['lq'].min()) and float(item['lq'].max()) <= 1.0
    batch = next(iter(fdata.build_dataloader(ds)))
    assert batch['lq'].shape == (1, 3, 8, 12) and batch['lq_path'][0].endswith('a.png')
    single = fdata.build_dataset(dict(type='SingleImageDataset', name='s', dataroot_lq=str(lq)))
    assert len(single) == 3 and 'gt' not in single[0]

    opt = dict(name='demo', model_type='FeMaSRModel', scale=4,
               datasets=dict(val=dict(name='d', type='PairedImageDataset', dataroot_lq=str(lq), dataroot_gt=str(gt),
                                      io_backend=dict(type='disk'))),
               network_g=dict(type='FeMaSRNet', gt_resolution=256, norm_type='gn', act_type='silu', scale_factor=4,
                              codebook_params=[[32, 1024, 512]], LQ_stage=True, frozen_module_keywords=['quantize']),
               path=dict(pretrain_network_g=None, strict_load=False), val=dict(save_img=True, suffix=None))
    p = tmp_path / 'opt.yml'
    p.write_text(yaml.safe_dump(opt))
    parsed = parse_options(str(p))
    assert parsed['is_train'] is False and parsed['datasets']['val']['phase'] == 'val'
    assert parsed['path']['visualization'].endswith(os.path.join('results', 'demo', 'visualization'))
    net = build_network(parsed['network_g'])                        # the YAML block goes through verbatim
    assert type(net).__name__ == 'FeMaSRNet' and net.scale_factor == 4


def test_metrics_definitions():
    rng = np.random.RandomState(1)
    a = rng.randint(0, 256, (40, 44, 3), dtype=np.uint8)
    b = a.copy()
    b[10:20, 10:20] = np.clip(b[10:20, 10:20].astype(int) + 9, 0, 255).astype(np.uint8)
    mse = np.mean((a[4:-4, 4:-4].astype(np.float64) - b[4:-4, 4:-4].astype(np.float64)) ** 2)
    assert abs(fm.calculate_psnr(a, b, crop_border=4) - 10 * np.log10(255 ** 2 / mse)) < 1e-9
    assert fm.calculate_psnr(a, a) == float('inf')
    assert abs(fm.calculate_ssim(a, a, crop_border=4, test_y_channel=True) - 1.0) < 1e-12
    s = fm.calculate_ssim(a, b, crop_border=4, test_y_channel=True)
    assert 0.0 < s < 1.0
    y = fm._to_y(np.full((2, 2, 3), 255, dtype=np.uint8))
    assert np.allclose(y, 235.0)                                    # BT.601 studio range
    t = torch.tensor([[[[-0.2, 0.5], [1.3, 0.00196]]]]).repeat(1, 3, 1, 1)
    img = fm.tensor2img(t)
    assert img.shape == (2, 2, 3) and img.dtype == np.uint8 and img[0, 0, 0] == 0 and img[1, 0, 0] == 255
    assert img[0, 1, 0] == 128 and img[1, 1, 0] == 0                # 127.5 -> 128 (half to even), 0.4998 -> 0
