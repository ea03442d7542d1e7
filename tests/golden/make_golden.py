"""Generate the committed golden vectors by running the REFERENCE itself.

Runs only in the development container (needs /root/reference; see
tools/ref_harness.py).  Usage:  python tests/golden/make_golden.py
Writes tests/golden/*.npz — DATA only: seeds/inputs, reference outputs, VQ index
maps, the reference's own best/runner-up VQ distances (for the documented
near-tie rule) and per-stage probes.  Weights are NOT stored: both sides
regenerate them from femasr_amd.synth (seed + state-dict key).

Reference entry points exercised (all under /root/reference/basicsr/archs):
  femasr_arch.py:449-468 FeMaSRNet.test        femasr_arch.py:387-447 test_tile
  femasr_arch.py:470-479 forward (HQ)          femasr_arch.py:376-385 decode_indices
  network_swinir.py:164-279 SwinTransformerBlock (unit)   femasr_arch.py:14-112 VectorQuantizer (unit)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from femasr_amd import synth  # noqa: E402
from ref_harness import import_reference_arch, import_reference_swin  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PROBE_CAP = 1 << 12          # elements kept per probe (deterministic subsample above this)


def subsample(name, arr, seed=1234):
    flat = np.ascontiguousarray(arr).reshape(-1)
    if flat.size <= PROBE_CAP:
        return np.arange(flat.size, dtype=np.int32), flat.copy()
    pos = (synth.uniform01(seed, 'probe.' + name, PROBE_CAP) * flat.size).astype(np.int32)
    return pos, flat[pos].copy()


def build_ref(cfg, seed, codebook):
    arch = import_reference_arch()
    net = arch.FeMaSRNet(**cfg).eval()
    w = synth.fill_state_dict(net.state_dict(), seed, codebook)
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not missing.unexpected_keys
    return net


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().numpy()


def run_with_probes(net, fn, x):
    """Run fn(x) capturing stage outputs (NHWC) and the VQ internals."""
    probes, vqinfo, hooks = {}, {}, []

    def hook(name, tok_hw=None):
        def _h(mod, inp, out):
            probes[name] = nhwc(out)
        return _h

    enc = net.multiscale_encoder
    hooks.append(enc.in_conv.register_forward_hook(hook('in_conv')))
    for i, blk in enumerate(enc.blocks):
        hooks.append(blk.register_forward_hook(hook(f'enc_block{i}')))
    hooks.append(net.before_quant_group[0].register_forward_hook(hook('z')))
    hooks.append(net.after_quant_group[0].register_forward_hook(hook('after_quant')))

    def vq_hook(mod, inp, out):
        z = inp[0].detach()
        probes['z_q'] = nhwc(out[0])
        zf = z.permute(0, 2, 3, 1).reshape(-1, mod.e_dim)
        d = mod.dist(zf, mod.embedding.weight.detach())
        top2 = torch.topk(d, 2, dim=1, largest=False)
        vqinfo['indices'] = out[2].detach().numpy().astype(np.int64)
        vqinfo['d_best'] = top2.values[:, 0].numpy()
        vqinfo['d_second'] = top2.values[:, 1].numpy()
        vqinfo['idx_second'] = top2.indices[:, 1].numpy().astype(np.int64)
    hooks.append(net.quantize_group[0].register_forward_hook(vq_hook))
    for i, blk in enumerate(net.decoder_group):
        hooks.append(blk.register_forward_hook(hook(f'dec{i}')))
    with torch.no_grad():
        y = fn(torch.from_numpy(x))
    for h in hooks:
        h.remove()
    return y, probes, vqinfo


def save_net_case(name, cfg, seed, codebook, in_shape, mode, out_stride=1, **kw):
    net = build_ref(cfg, seed, codebook)
    x = synth.synth_input(seed + 100, in_shape)
    if mode == 'test':
        y, probes, vq = run_with_probes(net, net.test, x)
    elif mode == 'forward':
        y, probes, vq = run_with_probes(net, lambda t: net(t)[0], x)
    elif mode == 'test_tile':
        with torch.no_grad():
            y = net.test_tile(torch.from_numpy(x), kw['tile_size'], kw['tile_pad'])
        probes, vq = {}, {}
    else:
        raise ValueError(mode)
    y = y.numpy()
    rec = {
        'cfg_LQ_stage': np.array(int(cfg.get('LQ_stage', False))),
        'cfg_scale_factor': np.array(int(cfg.get('scale_factor', 4))),
        'seed': np.array(seed), 'input_seed': np.array(seed + 100),
        'codebook': np.array(codebook), 'mode': np.array(mode),
        'in_shape': np.array(in_shape), 'out_shape': np.array(y.shape),
        'out_stride': np.array(out_stride),
        'output': y[:, :, ::out_stride, ::out_stride].copy(),
        'out_mean': np.array(y.astype(np.float64).mean()), 'out_absmax': np.array(np.abs(y).max()),
    }
    for k, v in kw.items():
        rec['kw_' + k] = np.array(v)
    for k, v in vq.items():
        rec['vq_' + k] = v
    for k, v in probes.items():
        pos, val = subsample(name + '.' + k, v)
        rec['probe_pos_' + k] = pos
        rec['probe_val_' + k] = val
        rec['probe_shape_' + k] = np.array(v.shape)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    print(f'{name}: out {y.shape} absmax {np.abs(y).max():.3f}', {k: v.shape for k, v in vq.items() if k == 'indices'})


def save_decode_indices_case():
    cfg = dict(codebook_params=[[32, 1024, 512]], LQ_stage=False)
    net = build_ref(cfg, 7, 'trained')
    idx = (synth.uniform01(7, 'decode.indices', 1 * 6 * 10) * 1024).astype(np.int64).reshape(1, 1, 6, 10)
    with torch.no_grad():
        y = net.decode_indices(torch.from_numpy(idx)).numpy()
    np.savez_compressed(os.path.join(OUT, 'hq_decode_indices.npz'), seed=np.array(7), codebook=np.array('trained'),
                        indices=idx, output=y)
    print('hq_decode_indices', y.shape)


def save_swin_block_case():
    """One shifted + one unshifted SwinTransformerBlock at a non-(32,32) resolution (mask recomputed path)."""
    swin = import_reference_swin()
    rec = {}
    for shift in (0, 4):
        blk = swin.SwinTransformerBlock(dim=256, input_resolution=(32, 32), num_heads=8, window_size=8,
                                        shift_size=shift).eval()
        sd = blk.state_dict()
        w = synth.fill_state_dict(sd, 11 + shift)
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
        x = synth.uniform(21 + shift, 'swin.x', (2, 16 * 24, 256), -2.0, 2.0)
        with torch.no_grad():
            y = blk(torch.from_numpy(x), (16, 24)).numpy()
        rec[f'y_shift{shift}'] = y
        rec[f'seed_shift{shift}'] = np.array(11 + shift)
        rec[f'xseed_shift{shift}'] = np.array(21 + shift)
    np.savez_compressed(os.path.join(OUT, 'unit_swin_block.npz'), **rec)
    print('unit_swin_block ok')


def save_vq_case():
    """VectorQuantizer with a crafted EXACT tie (duplicated codebook rows -> first index must win)."""
    arch = import_reference_arch()
    vq = arch.VectorQuantizer(1024, 512).eval()
    cb = synth.uniform(5, 'vq.codebook', (1024, 512), -1.0, 1.0)
    cb[700] = cb[13]            # exact duplicates: argmin must return 13 for vectors nearest to it
    cb[901] = cb[13]
    z = synth.uniform(6, 'vq.z', (2, 512, 5, 7), -1.0, 1.0)
    z[0, :, 0, 0] = cb[13] * 0.97           # nearest code = row 13 (tie with 700, 901)
    z[1, :, 4, 6] = cb[700] * 1.02
    vq.embedding.weight.data.copy_(torch.from_numpy(cb))
    with torch.no_grad():
        zq, _, idx = vq(torch.from_numpy(z))
    np.savez_compressed(os.path.join(OUT, 'unit_vq_tie.npz'), indices=idx.numpy(), z_q=zq.numpy())
    assert idx[0, 0, 0, 0] == 13 and idx[1, 0, 4, 6] == 13
    print('unit_vq_tie ok', idx[0, 0, 0, 0].item(), idx[1, 0, 4, 6].item())


def save_state_dict_keys():
    """Key / shape / dtype list of the reference state dict per config (what checkpoints contain)."""
    import json
    arch = import_reference_arch()
    cfgs = {'x4': dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4),
            'x2': dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=2),
            'hq': dict(codebook_params=[[32, 1024, 512]], LQ_stage=False)}
    out = {}
    for name, cfg in cfgs.items():
        sd = arch.FeMaSRNet(**cfg).state_dict()
        out[name] = [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
    with open(os.path.join(OUT, 'state_dict_keys.json'), 'w') as f:
        json.dump(out, f)
    print('state_dict_keys', {k: len(v) for k, v in out.items()})


def main():
    torch.set_num_threads(8)
    save_state_dict_keys()
    lq4 = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
    lq2 = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=2)
    hq = dict(codebook_params=[[32, 1024, 512]], LQ_stage=False)
    save_net_case('x4_small_init', lq4, 0, 'init', (2, 3, 20, 28), 'test')
    save_net_case('x4_small_trained', lq4, 1, 'trained', (1, 3, 24, 40), 'test')
    save_net_case('x2_small_trained', lq2, 2, 'trained', (1, 3, 40, 24), 'test')
    save_net_case('hq_small_trained', hq, 3, 'trained', (1, 3, 64, 48), 'forward')
    save_net_case('x4_tiled_trained', lq4, 1, 'trained', (1, 3, 40, 56), 'test_tile', tile_size=24, tile_pad=8)
    save_net_case('x4_tile128_init', lq4, 0, 'init', (1, 3, 128, 128), 'test', out_stride=4)
    save_net_case('x4_tile128_trained', lq4, 1, 'trained', (1, 3, 128, 128), 'test', out_stride=4)
    save_decode_indices_case()
    save_swin_block_case()
    save_vq_case()


if __name__ == '__main__':
    main()
