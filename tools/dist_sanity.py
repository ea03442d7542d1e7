"""RCCL sanity on one GPU box: run under torchrun (any world size); checks that the tile-parallel path (one all_gather
over the NCCL/RCCL backend) reproduces the single-process test_tile output bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from femasr_amd import distributed as fd, synth  # noqa: E402
from femasr_amd.archs.femasr_arch import FeMaSRNet  # noqa: E402


def main():
    rank, world, local = fd.init_from_env()
    torch.cuda.set_device(local)
    if not torch.distributed.is_initialized():       # world size 1: still go through RCCL (one-rank group)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        torch.distributed.init_process_group(backend='nccl', rank=0, world_size=1)
    dev = torch.device('cuda', local)
    net = FeMaSRNet(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
    w = synth.fill_state_dict(net.state_dict(), 5, 'trained')
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    net = net.to(dev).eval()
    x = torch.from_numpy(synth.synth_input(7, (1, 3, 200, 168))).to(dev)
    ref = net.test_tile(x, 96, 16)
    got = fd.test_tile_parallel(net, x, 96, 16)
    same = bool(torch.equal(ref, got))
    print(f'rank {rank}/{world}: tile-parallel == test_tile: {same}; backend {torch.distributed.get_backend()}', flush=True)
    torch.distributed.barrier()
    if not same:
        raise SystemExit(1)


if __name__ == '__main__':
    main()
