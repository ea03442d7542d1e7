"""Validation-time counterparts of the reference's folder datasets (basicsr/data/paired_image_dataset.py:56-113,
single_image_dataset.py) - 'folder' mode, `io_backend: disk`, phase != 'train' only.

Each item is {'lq', ['gt'], 'lq_path', ['gt_path']} with float32 CHW RGB tensors in [0,1] (`img2tensor(bgr2rgb=True)`
of `cv2.imread(...)/255.`).  Files are paired by SORTED path (the reference pairs by `os.walk` order, which is the
same listing for both folders on one filesystem but is not sorted).  `crop_eval_size` (paired_image_dataset.py:97-104):
a paired RANDOM crop of the gt to that size (lq to size // scale), positions drawn with `random.randint` like the
reference's `paired_random_crop` (transforms.py:26-90)."""
import os
import random

import numpy as np
import torch

_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.tif', '.tiff', '.webp')


def make_dataset(folder):
    assert os.path.isdir(folder), f'{folder} is not a valid directory'
    out = []
    for root, _, fnames in sorted(os.walk(folder, followlinks=True)):
        out += [os.path.join(root, f) for f in sorted(fnames) if f.lower().endswith(_EXT)]
    return out


def _read(path):
    from PIL import Image
    return torch.from_numpy(np.asarray(Image.open(path).convert('RGB'), dtype=np.float32).transpose(2, 0, 1) / 255.0)


class PairedImageDataset(torch.utils.data.Dataset):
    def __init__(self, opt):
        self.opt = opt
        self.lq_paths, self.gt_paths = make_dataset(opt['dataroot_lq']), make_dataset(opt['dataroot_gt'])
        assert len(self.lq_paths) == len(self.gt_paths), 'lq / gt folders have different numbers of images'

    def __len__(self):
        return len(self.gt_paths)

    def __getitem__(self, i):
        lq, gt = _read(self.lq_paths[i]), _read(self.gt_paths[i])
        size = self.opt.get('crop_eval_size', None)
        if size:
            scale = gt.shape[1] // lq.shape[1]
            ps = size // scale
            if gt.shape[1] != lq.shape[1] * scale or gt.shape[2] != lq.shape[2] * scale:
                raise ValueError(f'Scale mismatches. GT {tuple(gt.shape[1:])} is not {scale}x multiplication of LQ {tuple(lq.shape[1:])}.')
            if lq.shape[1] < ps or lq.shape[2] < ps:
                raise ValueError(f'LQ {tuple(lq.shape[1:])} is smaller than patch size ({ps}, {ps}). Please remove {self.gt_paths[i]}.')
            top, left = random.randint(0, lq.shape[1] - ps), random.randint(0, lq.shape[2] - ps)
            lq = lq[:, top:top + ps, left:left + ps].contiguous()
            gt = gt[:, top * scale:top * scale + size, left * scale:left * scale + size].contiguous()
        return {'lq': lq, 'gt': gt, 'lq_path': self.lq_paths[i], 'gt_path': self.gt_paths[i]}


class SingleImageDataset(torch.utils.data.Dataset):
    def __init__(self, opt):
        self.opt = opt
        self.paths = make_dataset(opt['dataroot_lq'])

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        return {'lq': _read(self.paths[i]), 'lq_path': self.paths[i]}


_DATASETS = {'PairedImageDataset': PairedImageDataset, 'SingleImageDataset': SingleImageDataset}


def build_dataset(dataset_opt):
    return _DATASETS[dataset_opt['type']](dataset_opt)


def build_dataloader(dataset, dataset_opt=None, **_):
    """Validation loader of the reference: batch size 1, no shuffle (basicsr/data/__init__.py, phase 'val'/'test')."""
    return torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=0)
