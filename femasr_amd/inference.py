#!/usr/bin/env python
"""CLI counterpart of the reference's `inference_femasr.py` (lines 19-69) on the MI355X path.

Same flags (-i -w -o -s --suffix --max_size), same per-image policy (sorted glob; whole-image `test()` when
h*w < max_size**2, else `test_tile()`), same output naming.  Differences, all forced by the environment:
  * images are read/written with PIL (cv2 is not installed); channel order is handled explicitly, PNG is exact;
  * weights are never downloaded (no network): pass -w, or --synthetic-seed N for the deterministic generator;
  * pre/post-processing runs on the GPU (femasr_amd.imgproc) — uint8 crosses PCIe, not fp32;
  * with torchrun (WORLD_SIZE > 1) large images are tile-sharded over the ranks (femasr_amd.distributed).
"""
import argparse
import glob
import os

import numpy as np
import torch


def load_weights(model, path):
    """`torch.load(path)['params']` with strict=False like inference_femasr.py:40; `module.` prefixes stripped
    like basicsr/models/base_model.py:258-323."""
    sd = torch.load(path, map_location='cpu')
    for key in ('params', 'params_ema', 'state_dict'):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
            break
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=False)


def main(argv=None):
    from PIL import Image
    from femasr_amd import distributed as fd
    from femasr_amd import imgproc, synth
    from femasr_amd.archs.femasr_arch import FeMaSRNet

    ap = argparse.ArgumentParser(description='FeMaSR inference on MI355X')
    ap.add_argument('-i', '--input', type=str, default='inputs', help='Input image or folder')
    ap.add_argument('-w', '--weight', type=str, default=None, help='path for model weights')
    ap.add_argument('-o', '--output', type=str, default='results', help='Output folder')
    ap.add_argument('-s', '--out_scale', type=int, default=4, help='The final upsampling scale of the image')
    ap.add_argument('--suffix', type=str, default='', help='Suffix of the restored image')
    ap.add_argument('--max_size', type=int, default=600, help='Max image size for whole image inference, otherwise use tiled_test')
    ap.add_argument('--synthetic-seed', type=int, default=None, help='use deterministic synthetic weights (no checkpoint)')
    ap.add_argument('--tile_size', type=int, default=240)
    ap.add_argument('--tile_pad', type=int, default=16)
    ap.add_argument('--streams', type=int, default=3, help='sub-batch streams inside one batched forward of the tiled branch (3 measured fastest on MI355X)')
    ap.add_argument('--decoder-math', choices=['fp32', 'fp32_strict', 'fp32_direct', 'bf16x3'], default='fp32',
                    help="arithmetic of the convs behind the codebook lookup (FeMaSRNet.decoder_math); 'fp32_strict' is bit-identical to the CPU oracle")
    args = ap.parse_args(argv)

    if not torch.cuda.is_available():
        raise SystemExit('femasr_amd.inference needs a GPU (there is no CPU fallback)')
    rank, world, local = fd.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    model = FeMaSRNet(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=args.out_scale)
    if args.weight is not None:
        load_weights(model, args.weight)
    elif args.synthetic_seed is not None:
        w = synth.fill_state_dict(model.state_dict(), args.synthetic_seed, 'trained')
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    else:
        raise SystemExit('no network here: pass -w <weights.pth> (FeMaSR_SRX4/SRX2_model_g.pth) or --synthetic-seed N')
    model.decoder_math = args.decoder_math
    model.num_streams = args.streams          # tiled images run as batched test() calls: sub-batches on internal streams (bit-identical for any split)
    model = model.to(dev).eval()

    if rank == 0:
        os.makedirs(args.output, exist_ok=True)
    paths = [args.input] if os.path.isfile(args.input) else sorted(glob.glob(os.path.join(args.input, '*')))
    for path in paths:
        img_name = os.path.basename(path)
        rgb = np.array(Image.open(path).convert('RGB'))                   # cv2.imread + BGR2RGB == RGB
        xu8 = torch.from_numpy(rgb).to(dev)
        h, w = rgb.shape[:2]
        if h * w < args.max_size ** 2:
            # whole image: ONE native call, uint8 in -> uint8 out (decode fused into the pad kernel, tensor2img into the crop kernel)
            u8 = model.test_u8(xu8) if rank == 0 else None
        else:
            # tiled: uint8 tiles in, uint8 tiles out - crops, the all-gather over xGMI and the paste move one byte per value, no fp32 image
            # or canvas is ever held (FeMaSRNet.test_tile_u8); with several ranks only rank 0 pastes
            if world > 1:
                u8 = fd.test_tile_parallel(model, xu8, args.tile_size, args.tile_pad, root_only=True)
            else:
                u8 = model.test_tile_u8(xu8, args.tile_size, args.tile_pad)
        if rank == 0:
            Image.fromarray(u8.cpu().numpy(), 'RGB').save(os.path.join(args.output, img_name))
    if world > 1:
        torch.distributed.barrier()


if __name__ == '__main__':
    main()
