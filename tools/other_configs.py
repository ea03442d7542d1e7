"""Timing of the other BASELINE.json configs (they are parity-test cases, not bench lines): config 4 (x2, B=32 of 256^2)
and config 5 (HQ autoencode forward, B=8 of 512^2), default bench settings (2 streams, bf16x3) and exact fp32."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import gpu_utils as G  # noqa: E402
from femasr_amd import synth  # noqa: E402
from helpers import synth_weights  # noqa: E402


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device('cuda', 0)
    for cfg, shape, call, out_px in (('x2', (32, 3, 256, 256), 'test', 32 * 512 * 512), ('hq', (8, 3, 512, 512), 'forward', 8 * 512 * 512)):
        net = G.build_net(cfg, synth_weights(cfg, 0, 'trained'), dev)
        x = torch.from_numpy(synth.synth_input(7, shape)).to(dev)
        for math, streams in (('bf16x3', 2), ('fp32', 2)):
            net.decoder_math, net.num_streams = math, streams
            fn = (lambda: net.test(x)) if call == 'test' else (lambda: net(x))
            ms = timed(fn)
            print(f'{cfg} {shape} {call} {math} streams={streams}: {ms:.1f} ms = {out_px / 1e6 / ms * 1e3:.1f} output MPix/s', flush=True)
        del net
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
