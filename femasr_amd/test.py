#!/usr/bin/env python
"""Counterpart of the reference's `basicsr/test.py:11-40` (test_pipeline): YAML options -> datasets -> build_model ->
model.validation(...) on the MI355X path.

    python -m femasr_amd.test -opt options/test_x4.yml

The option file has the reference's layout (name, model_type, scale, datasets.<x>.{name,type,dataroot_lq,dataroot_gt,
io_backend}, network_g, path.{pretrain_network_g,pretrain_network_hq,strict_load}, val.{save_img,suffix,metrics}).
Results go to `results/<name>/visualization/<dataset>/` like the reference's `make_exp_dirs` layout."""
import argparse
import logging
import os

import yaml


def parse_options(path):
    with open(path) as f:
        opt = yaml.safe_load(f)
    opt['is_train'] = False
    for phase, ds in (opt.get('datasets') or {}).items():
        ds['phase'] = phase.split('_')[0]
        ds.setdefault('scale', opt.get('scale'))
        for k in ('dataroot_gt', 'dataroot_lq'):
            if ds.get(k):
                ds[k] = os.path.expanduser(ds[k])
    results_root = os.path.join(opt.get('root_path', os.getcwd()), 'results', opt['name'])
    opt.setdefault('path', {})
    opt['path'].setdefault('results_root', results_root)
    opt['path'].setdefault('log', results_root)
    opt['path'].setdefault('visualization', os.path.join(results_root, 'visualization'))
    return opt


def test_pipeline(opt_path):
    from femasr_amd.data import build_dataloader, build_dataset
    from femasr_amd.models import build_model
    opt = parse_options(opt_path)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s %(levelname)s: %(message)s')
    log = logging.getLogger('femasr_amd')
    loaders = []
    for _, dataset_opt in sorted(opt['datasets'].items()):
        if dataset_opt['phase'] == 'train':
            continue
        test_set = build_dataset(dataset_opt)
        log.info('Number of test images in %s: %d', dataset_opt['name'], len(test_set))
        loaders.append(build_dataloader(test_set, dataset_opt))
    model = build_model(opt)
    results = {}
    for loader in loaders:
        name = loader.dataset.opt['name']
        log.info('Testing %s...', name)
        results[name] = model.validation(loader, current_iter=opt['name'], tb_logger=None,
                                         save_img=(opt.get('val') or {}).get('save_img', False))
    return results


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True, help='Path to option YAML file.')
    print(test_pipeline(ap.parse_args().opt))
