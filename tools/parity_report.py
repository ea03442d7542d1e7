"""Max-abs error table of the HIP path against every reference golden, every decoder math mode (numbers quoted in DESIGN.md)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from femasr_amd import synth  # noqa: E402
from femasr_amd.archs import build_network  # noqa: E402
from helpers import CONFIGS, cfg_name_of, golden_cfg, load_golden, synth_weights, weights_from_arch  # noqa: E402


def run(name, r2):
    g = load_golden(name)
    if r2:
        cfg = golden_cfg(g)
        w = weights_from_arch(cfg, int(g['seed']), str(g['codebook']), str(g['variant']))
    else:
        cfg = CONFIGS[cfg_name_of(g)]
        w = synth_weights(cfg_name_of(g), int(g['seed']), str(g['codebook']))
    net = build_network(dict(type='FeMaSRNet', **cfg))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    net = net.cuda().eval()
    x = torch.from_numpy(synth.synth_input(int(g['input_seed']), tuple(g['in_shape']))).cuda()
    st = int(g['out_stride']) if 'out_stride' in g else 1
    res = {}
    for math in ('fp32_direct', 'fp32_strict', 'fp32', 'bf16x3', 'fp32+fp32_linears'):
        net.linear_math = 'fp32' if math.endswith('fp32_linears') else 'bf16_split'
        net.decoder_math = math.split('+')[0]
        if str(g['mode']) == 'test':
            y, idx = net.test_with_all_indices(x)
        else:
            y, _, _, idx = net(x)
        y = y.cpu().numpy()[:, :, ::st, ::st]
        res[math] = (float(np.abs(y - g['output']).max()), int((idx[0].cpu().numpy().reshape(-1) != g['vq_indices'].reshape(-1)).sum()), y)
    d = lambda a, b: float(np.abs(res[a][2] - res[b][2]).max())
    print(f'{name:22s} |out|max {float(np.abs(g["output"]).max()):8.3f}  vs reference: direct {res["fp32_direct"][0]:.2e} strict(F4x4) {res["fp32_strict"][0]:.2e} '
          f'fp32(default) {res["fp32"][0]:.2e} bf16x3 {res["bf16x3"][0]:.2e} default with fp32-chain linears {res["fp32+fp32_linears"][0]:.2e} '
          f'(idx mismatches: default {res["fp32"][1]}, fp32-chain linears {res["fp32+fp32_linears"][1]});  '
          f'strict vs direct {d("fp32_strict", "fp32_direct"):.2e}  default vs strict {d("fp32", "fp32_strict"):.2e}  bf16x3 vs strict {d("bf16x3", "fp32_strict"):.2e}', flush=True)


for n in ('x4_small_init', 'x4_small_trained', 'x2_small_trained', 'hq_small_trained', 'x4_tile128_init', 'x4_tile128_trained',
          'x2_tile256_trained', 'hq_full512_trained'):
    run(n, False)
for n in ('x4_small_torchinit', 'x4_small_unscaled', 'hq2_small_trained', 'x4mc_small_trained'):
    run(n, True)
