"""Debug: the OST_120 tiled golden through the HIP path in every decoder mode; where does the output leave the reference?"""
import io
import os
import sys

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from PIL import Image  # noqa: E402
import gpu_utils as G  # noqa: E402
from femasr_amd import imgproc  # noqa: E402
from helpers import load_golden, weights_from_arch  # noqa: E402

dev = torch.device('cuda', 0)
g = load_golden('png_OST_120_tiled')
cfg = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
w = weights_from_arch(cfg, int(g['seed']), 'trained')
rgb = np.asarray(Image.open(io.BytesIO(g['png'].tobytes())).convert('RGB'))
x = imgproc.u8_to_input(torch.from_numpy(np.ascontiguousarray(rgb)).to(dev))
ref = g['output_f32_stride8']
outs = {}
for math in ('fp32_direct', 'fp32_strict', 'fp32'):
    net = G.build_net('x4', w, dev, decoder_math=math)
    for streams in (1, 2):
        net.num_streams = streams
        y = net.test_tile(x, int(g['tile_size']), int(g['tile_pad']))
        yn = y.cpu().numpy()
        d = np.abs(yn[:, :, ::8, ::8] - ref)
        pos = np.unravel_index(np.argmax(d), d.shape)
        bad = np.argwhere(d.max(axis=(0, 1)) > 1e-3)
        print(f'{math} streams={streams}: max-abs vs reference {d.max():.3e} at (c,y,x)={pos[1]},{pos[2] * 8},{pos[3] * 8}; '
              f'{len(bad)} strided pixels > 1e-3; y range {bad[:, 0].min() * 8 if len(bad) else -1}..{bad[:, 0].max() * 8 if len(bad) else -1} '
              f'x range {bad[:, 1].min() * 8 if len(bad) else -1}..{bad[:, 1].max() * 8 if len(bad) else -1}', flush=True)
        outs[(math, streams)] = yn
    # the tile classes one by one, against fp32_direct
    for (h0, h1, w0, w1) in ((0, 256, 0, 256), (0, 256, 224, 496), (224, 496, 224, 496), (224, 496, 0, 256)):
        crop = x[:, :, h0:h1, w0:w1].contiguous()
        net.num_streams = 1
        t = net.test(crop).cpu().numpy()
        outs[(math, 'crop', h0, w0)] = t
        if math != 'fp32_direct':
            dd = np.abs(t - outs[('fp32_direct', 'crop', h0, w0)])
            pos = np.unravel_index(np.argmax(dd), dd.shape)
            print(f'   crop y{h0}:{h1} x{w0}:{w1} [{math}] vs fp32_direct: max-abs {dd.max():.3e} at {pos}', flush=True)
