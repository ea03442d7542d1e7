cd $GRAFT_REPO_ROOT; timeout 600 python - <<'PY' 2>&1 | tail -8
import os, sys, time, torch
sys.path.insert(0, 'tests')
import gpu_utils as G
from femasr_amd import synth
from helpers import synth_weights
dev = torch.device('cuda', 0)
net = G.build_net('x4', synth_weights('x4', 0, 'trained'), dev)
net.decoder_math = 'bf16x3'
for B in (1, 2, 4, 8, 16, 32):
    x = torch.from_numpy(synth.synth_input(3, (B, 3, 128, 128))).to(dev)
    for streams in (1, 2):
        net.num_streams = streams
        net.test(x); torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n): net.test(x)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        print(f'B={B} streams={streams}: {ms:.2f} ms/step, {ms/B:.2f} ms/tile, {B*512*512/1e6/ms*1e3:.1f} MPix/s', flush=True)
PY
