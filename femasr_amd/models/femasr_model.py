"""Inference surface of the reference's `FeMaSRModel` (basicsr/models/femasr_model.py) on the MI355X path.

What is mirrored (same names, argument meaning and behaviour):
  __init__(opt)            femasr_model.py:20-70   build_network(opt['network_g']); LQ stage: frozen HQ net from
                                                   path.pretrain_network_hq (also loaded non-strictly into net_g);
                                                   path.pretrain_network_g with path.strict_load
  load_network             base_model.py:291-323   {'params'|'params_ema'} key, 'module.' prefix stripping, and with
                                                   strict=False same-name/different-size tensors are skipped
  feed_data / test         femasr_model.py:135-138,217-227   (whole image below 8000x8000 pixels, else test_tile)
  validation               base_model.py:45-57 -> nondist_validation femasr_model.py:234-328: per image feed/test,
                                                   tensor2img, save under path.visualization/<dataset>/<name>_<suffix|opt name>.png,
                                                   metric averages
  extract_gt_indices       femasr_model.py:144-146  `net_hq(gt)` -> indices
What is NOT here: training (optimizers, losses, discriminator, schedulers), best-model bookkeeping, tensorboard.
Metrics: the reference evaluates them with `pyiqa` (not installed here): 'psnr' and 'ssim' follow the BasicSR
definitions (crop_border, test_y_channel on the BT.601 Y of the uint8-rounded image); other types (lpips, ...) are
reported as skipped.  There is no CPU path: the module needs a GPU and the HIP library.
"""
import logging
import os
from collections import OrderedDict
from copy import deepcopy

import numpy as np
import torch

from .. import imgproc
from ..archs import build_network
from . import MODEL_REGISTRY

logger = logging.getLogger('femasr_amd')


def tensor2img(tensor, rgb2bgr=False, min_max=(0, 1)):
    """img_util.py:38-94 for the (1,3,H,W) / (3,H,W) case: clamp, scale, round half to even, uint8 HWC.
    The reference returns BGR for cv2.imwrite; images are written with PIL here, so the default stays RGB."""
    t = tensor.detach().float().cpu()
    if t.dim() == 4:
        t = t.squeeze(0)
    t = (t.clamp(*min_max) - min_max[0]) / (min_max[1] - min_max[0])
    img = (t.numpy().transpose(1, 2, 0) * 255.0).round().astype(np.uint8)
    return img[:, :, ::-1].copy() if rgb2bgr else img


def _to_y(img_u8_rgb):
    """BT.601 luma of an RGB uint8 image, the `rgb2ycbcr(..., y_only=True)` of BasicSR, range [16, 235]."""
    x = img_u8_rgb.astype(np.float64) / 255.0
    return (x @ np.array([65.481, 128.553, 24.966])) + 16.0


def calculate_psnr(img, img2, crop_border=0, test_y_channel=False, **_):
    a, b = img.astype(np.float64), img2.astype(np.float64)
    if crop_border:
        a, b = a[crop_border:-crop_border, crop_border:-crop_border], b[crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel:
        a, b = _to_y(a.astype(np.uint8)), _to_y(b.astype(np.uint8))
    mse = np.mean((a - b) ** 2)
    return float('inf') if mse == 0 else float(10.0 * np.log10(255.0 * 255.0 / mse))


def _ssim_plane(a, b):
    from scipy.signal import convolve2d
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    win = np.outer(g, g)

    def f(x):
        return convolve2d(x, win, mode='valid')
    mu1, mu2 = f(a), f(b)
    s1, s2, s12 = f(a * a) - mu1 * mu1, f(b * b) - mu2 * mu2, f(a * b) - mu1 * mu2
    return float((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))).mean())


def calculate_ssim(img, img2, crop_border=0, test_y_channel=False, **_):
    a, b = img.astype(np.float64), img2.astype(np.float64)
    if crop_border:
        a, b = a[crop_border:-crop_border, crop_border:-crop_border], b[crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel:
        return _ssim_plane(_to_y(a.astype(np.uint8)), _to_y(b.astype(np.uint8)))
    return float(np.mean([_ssim_plane(a[..., c], b[..., c]) for c in range(a.shape[2])]))


_METRICS = {'psnr': calculate_psnr, 'ssim': calculate_ssim}


@MODEL_REGISTRY.register()
class FeMaSRModel:
    def __init__(self, opt):
        self.opt = opt
        if opt.get('is_train', False):
            raise NotImplementedError('femasr_amd.models.FeMaSRModel is inference-only (is_train must be false)')
        if not torch.cuda.is_available():
            raise RuntimeError('FeMaSRModel needs a GPU: the hot path has no CPU implementation')
        self.device = torch.device('cuda', int(opt.get('local_rank', 0)))
        self.is_train = False
        path = opt.get('path', {}) or {}
        self.net_g = build_network(opt['network_g']).to(self.device).eval()

        self.LQ_stage = opt['network_g'].get('LQ_stage', False)
        self.net_hq = None
        if self.LQ_stage:
            load_path = path.get('pretrain_network_hq', None)
            if load_path is not None:       # the reference asserts it (needed for training); inference can do without
                hq_opt = deepcopy(opt['network_g'])
                hq_opt['LQ_stage'] = False
                self.net_hq = build_network(hq_opt).to(self.device).eval()
                self.load_network(self.net_hq, load_path, path.get('strict_load', True))
                self.load_network(self.net_g, load_path, False)
        load_path = path.get('pretrain_network_g', None)
        if load_path is not None:
            logger.info('Loading net_g from %s', load_path)
            self.load_network(self.net_g, load_path, path.get('strict_load', True))
        self.metric_results = {}

    # ---- base_model.py:258-323
    def load_network(self, net, load_path, strict=True, param_key='params'):
        if str(load_path).startswith('https://'):
            raise RuntimeError(f'no network in this environment: download {load_path} and pass the local file')
        load_net = torch.load(load_path, map_location='cpu')
        if param_key is not None:
            if param_key not in load_net and 'params' in load_net:
                param_key = 'params'
            load_net = load_net[param_key]
        load_net = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in load_net.items())
        crt = net.state_dict()
        missing, unexpected = sorted(set(crt) - set(load_net)), sorted(set(load_net) - set(crt))
        if missing or unexpected:
            logger.warning('Current net - loaded net: %s; loaded net - current net: %s', missing, unexpected)
        if not strict:
            for k in sorted(set(crt) & set(load_net)):
                if tuple(crt[k].shape) != tuple(load_net[k].shape):
                    logger.warning('Size different, ignore [%s]: crt_net %s; load_net %s', k, tuple(crt[k].shape),
                                   tuple(load_net[k].shape))
                    load_net[k + '.ignore'] = load_net.pop(k)
        return net.load_state_dict(load_net, strict=strict)

    def feed_data(self, data):
        self.lq = data['lq'].to(self.device)
        if 'gt' in data:
            self.gt = data['gt'].to(self.device)
        elif hasattr(self, 'gt'):
            del self.gt             # never score an image against the previous item's ground truth

    @torch.no_grad()
    def test(self):
        min_size = 8000 * 8000
        _, _, h, w = self.lq.shape
        self.output = self.net_g.test(self.lq) if h * w < min_size else self.net_g.test_tile(self.lq)

    @torch.no_grad()
    def extract_gt_indices(self, gt=None):
        """`self.gt_rec, _, _, gt_indices = self.net_hq(self.gt)` (femasr_model.py:145-146) as an inference service."""
        if self.net_hq is None:
            raise RuntimeError('no HQ network: set path.pretrain_network_hq in an LQ-stage option file')
        self.gt_rec, _, _, gt_indices = self.net_hq(self.gt if gt is None else gt.to(self.device))
        return gt_indices

    def validation(self, dataloader, current_iter, tb_logger, save_img=False, save_as_dir=None):
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img, save_as_dir)

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img, save_as_dir=None):
        from PIL import Image
        dataset_name = dataloader.dataset.opt['name']
        val_opt = self.opt.get('val', {}) or {}
        metrics = val_opt.get('metrics') or {}
        self.metric_results = {name: 0.0 for name in metrics}
        skipped = sorted(name for name, m in metrics.items() if m.get('type') not in _METRICS)
        if skipped:
            logger.warning('metrics %s need pyiqa (not installed): skipped', skipped)
        n = 0
        for val_data in dataloader:
            img_name = os.path.splitext(os.path.basename(val_data['lq_path'][0]))[0]
            self.feed_data(val_data)
            self.test()
            # tensor2img (img_util.py:38-94) on the GPU: clamp / x255 / round-half-even in a HIP kernel, uint8 crosses PCIe
            sr_img = imgproc.output_to_u8(self.output).cpu().numpy()
            if save_img:
                suffix = val_opt.get('suffix') or self.opt['name']
                save_img_path = os.path.join(self.opt['path']['visualization'], dataset_name, f'{img_name}_{suffix}.png')
                os.makedirs(os.path.dirname(save_img_path), exist_ok=True)
                Image.fromarray(sr_img, 'RGB').save(save_img_path)
                if save_as_dir:
                    os.makedirs(save_as_dir, exist_ok=True)
                    Image.fromarray(sr_img, 'RGB').save(os.path.join(save_as_dir, f'{img_name}.png'))
            if metrics and hasattr(self, 'gt'):
                gt_img = imgproc.output_to_u8(self.gt).cpu().numpy()
                for name, m in metrics.items():
                    fn = _METRICS.get(m.get('type'))
                    if fn is not None:
                        self.metric_results[name] += fn(sr_img, gt_img, **{k: v for k, v in m.items() if k not in ('type', 'better')})
            del self.lq, self.output
            if hasattr(self, 'gt'):
                del self.gt
            n += 1
        for name in self.metric_results:
            self.metric_results[name] = self.metric_results[name] / max(n, 1) if name not in skipped else None
        if metrics:
            logger.info('Validation %s: %s', dataset_name, self.metric_results)
        return self.metric_results

    def get_current_visuals(self):
        out = OrderedDict(lq=self.lq.detach().cpu(), result=self.output.detach().cpu())
        if hasattr(self, 'gt'):
            out['gt'] = self.gt.detach().cpu()
        return out
