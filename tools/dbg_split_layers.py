"""Debugging aid (GPU box): every 1x1 / Linear layer of one fixture's oracle forward is re-run through femasr_conv2d(w_bf16s) on the
oracle's own inputs; the first outputs that differ are dumped (operands of that row / column) to gpurun_out/ for offline analysis."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gpu_utils as G                      # noqa: E402
from femasr_amd import synth               # noqa: E402
from helpers import cfg_name_of, load_golden, synth_weights      # noqa: E402
from oracle import oracle as orc           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'x4_small_init'
g = load_golden(name)
cn = cfg_name_of(g)
w = synth_weights(cn, int(g['seed']), str(g['codebook']))
x = synth.synth_input(int(g['input_seed']), tuple(g['in_shape']))
calls = []
real = orc.linear_bf16s


def spy(x_rows, w_oi, bias, act=0, res1=None, res2=None, scalar=False):
    y = real(x_rows, w_oi, bias, act, res1, res2, scalar)
    calls.append((np.array(x_rows), np.array(w_oi), np.array(bias), act, None if res1 is None else np.array(res1), y))
    return y


orc.linear_bf16s = spy
from helpers import oracle_net             # noqa: E402
oracle_net(cn, w).test(x)
print(len(calls), 'split-linear calls')
dumped = 0
for li, (xr, woi, b, act, r1, yo) in enumerate(calls):
    rows, cin = xr.shape
    cout = woi.shape[0]
    y = G.conv2d(xr.reshape(1, rows, 1, cin), np.ascontiguousarray(woi.T).reshape(1, 1, cin, cout), b, 1, act=act,
                 res1=None if r1 is None else r1.reshape(1, rows, 1, cout), bf16s=True).reshape(rows, cout)
    bad = np.argwhere(y != yo)
    if len(bad):
        print(f'layer {li}: rows {rows} cin {cin} cout {cout} act {act} res {r1 is not None}: {len(bad)} outputs differ, max-abs {np.abs(y - yo).max():.3e}')
        if dumped < 3:
            r, c = bad[0]
            np.savez(os.path.join(ROOT, 'gpurun_out', f'split_mismatch_{dumped}.npz'), x=xr[r], w=woi[c], b=b[c], act=act,
                     res=0.0 if r1 is None else r1[r, c], y_gpu=y[r, c], y_oracle=yo[r, c], layer=li, row=r, col=c,
                     bad_rows=bad[:, 0], bad_cols=bad[:, 1])
            dumped += 1
print('done')
