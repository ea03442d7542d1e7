"""GPU end-to-end test of the YAML -> build_model -> validation pipeline: the PNGs the pipeline
saves must equal, bit for bit, the CPU ORACLE chain (uint8 -> /255 -> oracle test() -> clamp / x255 / round -> uint8), and
the metrics it reports must equal the metric functions evaluated on those oracle images."""
import os

import numpy as np
import pytest
import torch
import yaml

from helpers import CONFIGS, synth_weights

pytestmark = pytest.mark.gpu


def _png(path, arr):
    from PIL import Image
    Image.fromarray(arr, 'RGB').save(path)


def test_test_pipeline_end_to_end(cuda_device, tmp_path):
    import gpu_utils as G
    from femasr_amd import imgproc
    from femasr_amd.test import test_pipeline
    rng = np.random.RandomState(3)
    lq, gt = tmp_path / 'lq', tmp_path / 'gt'
    lq.mkdir(); gt.mkdir()
    imgs = {}
    for name, (h, w) in (('a.png', (24, 40)), ('b.png', (33, 17))):
        imgs[name] = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        _png(str(lq / name), imgs[name])
        _png(str(gt / name), rng.randint(0, 256, (4 * h, 4 * w, 3), dtype=np.uint8))
    weights = synth_weights('x4', 11, 'trained')
    ckpt = tmp_path / 'net_g.pth'        # released-checkpoint format: {'params': ...}, DataParallel 'module.' prefixes
    torch.save({'params': {'module.' + k: torch.from_numpy(v) for k, v in weights.items()}}, str(ckpt))
    opt = dict(name='pipe', model_type='FeMaSRModel', scale=4, root_path=str(tmp_path),
               datasets=dict(val=dict(name='tiny', type='PairedImageDataset', dataroot_lq=str(lq), dataroot_gt=str(gt),
                                      io_backend=dict(type='disk'))),
               network_g=dict(type='FeMaSRNet', gt_resolution=256, norm_type='gn', act_type='silu', scale_factor=4,
                              codebook_params=[[32, 1024, 512]], LQ_stage=True, decoder_math='fp32_strict',     # bit-exact PNGs below
                              frozen_module_keywords=['quantize', 'decoder', 'after_quant_group', 'out_conv']),
               path=dict(pretrain_network_g=str(ckpt), strict_load=False),
               val=dict(save_img=True, suffix=None,
                        metrics=dict(psnr=dict(type='psnr', crop_border=4, test_y_channel=True),
                                     ssim=dict(type='ssim', crop_border=4, test_y_channel=True),
                                     lpips=dict(type='lpips', better='lower'))))
    p = tmp_path / 'opt.yml'
    p.write_text(yaml.safe_dump(opt))
    results = test_pipeline(str(p))
    assert set(results) == {'tiny'}
    r = results['tiny']
    assert np.isfinite(r['psnr']) and 0.0 < r['ssim'] < 1.0 and r['lpips'] is None       # lpips needs pyiqa: skipped

    from PIL import Image
    from helpers import oracle_net
    from oracle import oracle as orc
    from femasr_amd.models.femasr_model import calculate_psnr, calculate_ssim
    onet = oracle_net('x4', weights)
    psnr = ssim = 0.0
    for name, arr in imgs.items():
        want = orc.image_f32_to_u8(onet.test(orc.image_u8_to_f32(arr)))         # the oracle chain, CPU
        saved = tmp_path / 'results' / 'pipe' / 'visualization' / 'tiny' / (name[:-4] + '_pipe.png')
        got = np.asarray(Image.open(str(saved)).convert('RGB'))
        assert got.shape == want.shape == (arr.shape[0] * 4, arr.shape[1] * 4, 3)
        assert np.array_equal(got, want), name
        gt_img = np.asarray(Image.open(str(gt / name)).convert('RGB'))
        psnr += calculate_psnr(want, gt_img, crop_border=4, test_y_channel=True) / len(imgs)
        ssim += calculate_ssim(want, gt_img, crop_border=4, test_y_channel=True) / len(imgs)
    assert abs(r['psnr'] - psnr) < 1e-9 and abs(r['ssim'] - ssim) < 1e-12


def test_hq_index_extraction_service(cuda_device, tmp_path):
    """LQ-stage option with path.pretrain_network_hq: the frozen HQ net yields gt_indices (femasr_model.py:144-146) and
    `decode_indices` reproduces its reconstruction."""
    from femasr_amd.models import build_model
    hq_w = synth_weights('hq', 5, 'trained')
    ckpt = tmp_path / 'hq.pth'
    torch.save({'params': {k: torch.from_numpy(v) for k, v in hq_w.items()}}, str(ckpt))
    opt = dict(name='hqsvc', model_type='FeMaSRModel', scale=4, is_train=False,
               network_g=dict(type='FeMaSRNet', **CONFIGS['x4']),
               path=dict(pretrain_network_hq=str(ckpt), pretrain_network_g=None, strict_load=False), val={})
    model = build_model(opt)
    assert model.net_hq is not None and not model.net_hq.LQ_stage
    gt = torch.from_numpy(np.random.RandomState(2).rand(1, 3, 64, 96).astype(np.float32))
    idx_list = model.extract_gt_indices(gt)          # a list with one entry per codebook, as the reference returns it
    assert isinstance(idx_list, list) and len(idx_list) == 1
    idx = idx_list[0]
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (1, 1, 8, 12)
    # the indices are the ORACLE's (HQ forward of the reference restatement), not merely self-consistent
    from helpers import oracle_net
    yo, io = oracle_net('hq', hq_w).forward(gt.numpy())
    assert np.array_equal(idx.cpu().numpy(), io)
    assert float(np.abs(model.gt_rec.cpu().numpy() - yo).max()) < 1e-4
    rec = model.net_hq.decode_indices(idx)
    assert float((rec - model.gt_rec).abs().max()) <= 1e-5
    # forward(input, gt_indices) - what FeMaSRModel.optimize_parameters passes (femasr_model.py:145-146): gt_indices only changes
    # the loss in the reference (femasr_arch.py:69-91), so image and indices equal forward(input) and the losses are zeros
    lq = torch.from_numpy(np.random.RandomState(4).rand(1, 3, 32, 48).astype(np.float32)).cuda()      # forward(): no pad, Swin map 16x24
    out0, cl0, sl0, i0 = model.net_g(lq)
    out1, cl1, sl1, i1 = model.net_g(lq, gt_indices=idx_list)
    assert torch.equal(out0, out1) and torch.equal(i0[0], i1[0]) and float(cl1) == 0.0 and float(sl1) == 0.0
    with pytest.raises(ValueError):
        model.net_g(lq, gt_indices=[])
    # the HQ checkpoint was also loaded (non-strictly) into net_g: shared decoder / codebook tensors are identical
    g, h = model.net_g.state_dict(), model.net_hq.state_dict()
    k = 'quantize_group.0.embedding.weight'
    assert torch.equal(g[k].cpu(), h[k].cpu())
