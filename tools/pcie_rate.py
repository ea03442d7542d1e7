"""One-off measurement for DESIGN 6: the bench step WITH host<->device transfers (pinned host buffers): fp32 tiles in,
fp32 HR tiles out (what a host-buffer caller of the reference's interface pays) and uint8 in / uint8 out through the
GPU-side pre/post kernels (what the CLI path pays)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import gpu_utils as G  # noqa: E402
from femasr_amd import synth  # noqa: E402
from helpers import synth_weights  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    net = G.build_net('x4', synth_weights('x4', 0, 'trained'), dev)
    net.num_streams, net.decoder_math = 2, 'bf16x3'
    B = 16
    xh = torch.from_numpy(synth.synth_input(1000, (B, 3, 128, 128))).pin_memory()
    yh = torch.empty((B, 3, 512, 512), dtype=torch.float32).pin_memory()
    xd = xh.to(dev)

    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def resident():
        net.test(xd)

    def fp32_pcie():
        y = net.test(xh.to(dev, non_blocking=True))
        yh.copy_(y, non_blocking=True)

    xu8 = (xh * 255).round().to(torch.uint8).pin_memory()
    yu8 = torch.empty((B, 3, 512, 512), dtype=torch.uint8).pin_memory()

    def u8_pcie():
        x = xu8.to(dev, non_blocking=True).float() / 255.0
        y = net.test(x)
        yu8.copy_((y.clamp(0, 1) * 255.0).round().to(torch.uint8), non_blocking=True)

    mp = B * 512 * 512 / 1e6
    for name, fn in (('inputs resident in HBM', resident), ('fp32 over PCIe (3.1 MB in, 50 MB out)', fp32_pcie),
                     ('uint8 over PCIe (0.8 MB in, 12.6 MB out)', u8_pcie)):
        ms = timed(fn)
        print(f'{name}: {ms:.2f} ms/step = {mp / ms * 1e3:.1f} MPix/s', flush=True)


if __name__ == '__main__':
    main()
