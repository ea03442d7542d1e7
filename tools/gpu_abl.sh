cd $GRAFT_REPO_ROOT; timeout 600 python - <<'PY' 2>&1 | tail -4
import os, sys, time, torch
sys.path.insert(0, 'tests')
import gpu_utils as G
from femasr_amd import synth
from helpers import synth_weights
dev = torch.device('cuda', 0)
net = G.build_net('x4', synth_weights('x4', 0, 'trained'), dev)
net.num_streams, net.decoder_math = 2, 'bf16x3'
x = torch.from_numpy(synth.synth_input(3, (1, 3, 2048, 2048))).to(dev)
for ts, pad in ((128, 0), (240, 16)):
    y = net.test_tile(x, ts, pad); torch.cuda.synchronize()
    t0 = time.perf_counter(); y = net.test_tile(x, ts, pad); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'config 3: 2048x2048 LR -> {tuple(y.shape)} test_tile(tile={ts}, pad={pad}) on 1 GPU: {dt*1e3:.0f} ms = {y.shape[2]*y.shape[3]/1e6/dt:.1f} output MPix/s, finite={bool(torch.isfinite(y).all())}', flush=True)
    del y
PY
