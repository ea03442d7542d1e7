"""Round-2 golden vectors, again by running the REFERENCE itself (development container only; see make_golden.py).

  python tests/golden/make_golden_r2.py

Adds the parity regimes the first set did not cover:
  x4_small_torchinit   weights drawn from torch's default initialisers (femasr_amd.synth.torch_default_tensor), reference
                       codebook init U(+-1/n_e) (femasr_arch.py:33): the index-tie-prone regime of a fresh model
  x4_small_unscaled    synthetic weights WITHOUT the 1/16 scale on out_conv (outputs ~16x larger)
  hq2_small_trained    two codebooks [[32,1024,512],[64,512,256]], HQ stage forward(): CombineQuantBlock's resize + concat
                       branch (fema_utils.py:92-99, femasr_arch.py:288-299,333-336)
  x4mc_small_trained   two codebooks on the LQ (x4) stage, test()
  png_chip / png_comic1  two images of the reference's testset/ through the CLI's arithmetic (inference_femasr.py:50-67:
                       imread -> /255 -> test() -> tensor2img, basicsr/utils/img_util.py:38-94): the PNG file bytes (data)
                       and the reference's uint8 RGB output under synthetic weights
DATA only is written (inputs, outputs, index maps); weights are regenerated from seeds on both sides.
"""
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from femasr_amd import synth  # noqa: E402
from ref_harness import import_reference_arch  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def build_ref(cfg, seed, codebook, variant):
    arch = import_reference_arch()
    net = arch.FeMaSRNet(**cfg).eval()
    w = synth.fill_state_dict(net.state_dict(), seed, codebook, variant)
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not missing.unexpected_keys
    return net


def vq_record(net, q, rec, prefix):
    """Hook quantizer q: indices + the reference's own best / runner-up distances (near-tie rule)."""
    def hook(mod, inp, out):
        z = inp[0].detach()
        zf = z.permute(0, 2, 3, 1).reshape(-1, mod.e_dim)
        d = mod.dist(zf, mod.embedding.weight.detach())
        top2 = torch.topk(d, 2, dim=1, largest=False)
        rec[prefix + 'indices'] = out[2].detach().numpy().astype(np.int64)
        rec[prefix + 'd_best'] = top2.values[:, 0].numpy()
        rec[prefix + 'd_second'] = top2.values[:, 1].numpy()
        rec[prefix + 'idx_second'] = top2.indices[:, 1].numpy().astype(np.int64)
    return net.quantize_group[q].register_forward_hook(hook)


def save_case(name, cfg, seed, codebook, variant, in_shape, mode):
    net = build_ref(cfg, seed, codebook, variant)
    x = synth.synth_input(seed + 100, in_shape)
    rec = {}
    hooks = [vq_record(net, q, rec, 'vq_' if q == 0 else f'vq{q}_') for q in range(len(cfg['codebook_params']))]
    with torch.no_grad():
        y = (net.test(torch.from_numpy(x)) if mode == 'test' else net(torch.from_numpy(x))[0]).numpy()
    for h in hooks:
        h.remove()
    rec.update(cfg_LQ_stage=np.array(int(cfg.get('LQ_stage', False))), cfg_scale_factor=np.array(int(cfg.get('scale_factor', 4))),
               codebook_params=np.array(cfg['codebook_params']), seed=np.array(seed), input_seed=np.array(seed + 100),
               codebook=np.array(codebook), variant=np.array(variant), mode=np.array(mode), in_shape=np.array(in_shape),
               output=y, out_absmax=np.array(np.abs(y).max()))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    print(f'{name}: out {y.shape} absmax {np.abs(y).max():.4g}', {k: v.shape for k, v in rec.items() if k.endswith('indices')})


def save_png_case(name, fname, seed):
    """inference_femasr.py:50-67 on one testset image (cv2 is not installed: PNG decode is exact, so PIL gives the same
    pixels; BGR<->RGB swaps cancel between imread/img2tensor and tensor2img/imwrite)."""
    from PIL import Image
    raw = open(os.path.join('/root/reference/testset', fname), 'rb').read()
    rgb = np.asarray(Image.open(io.BytesIO(raw)).convert('RGB'))
    cfg = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
    net = build_ref(cfg, seed, 'trained', 'default')
    x = torch.from_numpy(rgb.transpose(2, 0, 1).astype(np.float32)).unsqueeze(0) / 255.
    rec = {}
    hk = vq_record(net, 0, rec, 'vq_')
    with torch.no_grad():
        y = net.test(x)
    hk.remove()
    # tensor2img (img_util.py:66-90) for one 3-channel image: clamp, (x-min)/(max-min), CHW->HWC, (x*255).round(), uint8
    t = y.squeeze(0).float().clone().clamp_(0, 1)
    out_u8 = (t.numpy().transpose(1, 2, 0) * 255.0).round().astype(np.uint8)
    rec.update(png=np.frombuffer(raw, np.uint8), seed=np.array(seed), codebook=np.array('trained'),
               output_f32_stride4=y.numpy()[:, :, ::4, ::4].copy(), out_absmax=np.array(float(y.abs().max())), output_u8=out_u8)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    print(f'{name}: {rgb.shape} -> {out_u8.shape}, float absmax {float(y.abs().max()):.3f}')


def main():
    torch.set_num_threads(8)
    lq4 = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
    save_case('x4_small_torchinit', lq4, 5, 'init', 'torchinit', (1, 3, 24, 40), 'test')
    save_case('x4_small_unscaled', lq4, 6, 'trained', 'unscaled', (1, 3, 24, 40), 'test')
    save_case('hq2_small_trained', dict(codebook_params=[[32, 1024, 512], [64, 512, 256]], LQ_stage=False), 8, 'trained', 'default',
              (1, 3, 64, 48), 'forward')
    save_case('x4mc_small_trained', dict(codebook_params=[[32, 1024, 512], [64, 512, 256]], LQ_stage=True, scale_factor=4), 9,
              'trained', 'default', (1, 3, 24, 40), 'test')
    save_png_case('png_chip', 'chip.png', 12)
    save_png_case('png_comic1', 'comic1.png', 12)


if __name__ == '__main__':
    main()
