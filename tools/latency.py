"""Small-batch latency of the hot path (x4, 128x128 tiles): ms per forward for B in 1..16, both math modes, 1 or 2 sub-batch
streams, eager launches vs one hipGraph replay per forward (FeMaSRNet.use_graph).  Prints one line per setting.
The reference's real test_tile loop is batch-1 (femasr_arch.py:405-429)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from femasr_amd import synth  # noqa: E402
from femasr_amd.archs import build_network  # noqa: E402


def main():
    quick = '--quick' in sys.argv[1:]            # fp32 only, eager launches only
    dev = torch.device('cuda', 0)
    net = build_network(dict(type='FeMaSRNet', codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4))
    sd = synth.fill_state_dict(net.state_dict(), 0, 'trained')
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.to(dev).eval()
    for math in (('fp32',) if quick else ('fp32', 'bf16x3')):
        net.decoder_math = math
        for B in (1, 2, 4, 8, 16):
            x = torch.from_numpy(synth.synth_input(3, (B, 3, 128, 128))).to(dev)
            for streams in ((1,) if B == 1 else (1, 2)):
                for graph in ((False,) if quick else (False, True)):
                    net.num_streams = streams
                    net.use_graph = graph
                    for _ in range(3):
                        net.test(x)
                    torch.cuda.synchronize()
                    n = 10
                    t0 = time.perf_counter()
                    for _ in range(n):
                        net.test(x)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / n * 1e3
                    # host time of one call (no sync): how far the CPU runs ahead
                    t0 = time.perf_counter()
                    net.test(x)
                    host = (time.perf_counter() - t0) * 1e3
                    torch.cuda.synchronize()
                    print(f'{math:7s} B={B:2d} streams={streams} graph={int(graph)}: {ms:8.3f} ms/forward  {ms / B:7.3f} ms/tile  '
                          f'{B * 0.262144 / ms * 1e3:6.2f} MPix/s  host {host:6.3f} ms', flush=True)
    net.use_graph = False


if __name__ == '__main__':
    main()
