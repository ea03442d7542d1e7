"""world_size-2 gloo test of the tile-parallel path (partition -> per-rank compute -> ONE all-gather ->
paste).  The network itself needs a GPU, so `test()` is replaced by a deterministic per-tile function;
what is checked is the host logic around it: every rank reconstructs exactly the single-rank result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from femasr_amd import distributed as fd
from femasr_amd.archs import build_network
from helpers import CONFIGS


def _fake_test(self, t):
    # depends on every pixel of the tile (max: exact, batch-size independent) and on position -> any mis-paste / mis-order is visible
    up = F.interpolate(t, scale_factor=4, mode='nearest')
    return up * 0.5 + t.amax(dim=(1, 2, 3), keepdim=True) + 0.001 * t.shape[2] + 0.01 * t.shape[3]


def _make_net():
    net = build_network(dict(type='FeMaSRNet', **CONFIGS['x4']))
    net.test = _fake_test.__get__(net)
    net.max_tile_batch = 3
    return net


def _worker(rank, world, port, x, ts, pad, expect, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = fd.init_from_env('gloo')
    assert (r, w) == (rank, world)
    y = fd.test_tile_parallel(_make_net(), x, ts, pad)
    q.put((rank, bool(torch.equal(y, expect))))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('shape,ts,pad', [((1, 3, 70, 100), 32, 8), ((2, 3, 64, 64), 32, 0)])
def test_two_rank_tile_parallel_equals_single_rank(shape, ts, pad):
    torch.manual_seed(0)
    x = torch.rand(shape)
    net = _make_net()
    expect = net.test_tile(x, ts, pad)
    # single-rank result itself equals the literal sequential loop of the reference
    seq = torch.zeros_like(expect)
    from femasr_amd import tiling
    for t in tiling.enumerate_tiles(shape[2], shape[3], ts, pad):
        o = net.test(x[:, :, t.y0p:t.y1p, t.x0p:t.x1p])
        ys, ye, xs, xe = t.out_src(4)
        a, b, c, d = t.out_dst(4)
        seq[:, :, a:b, c:d] = o[:, :, ys:ye, xs:xe]
    assert torch.equal(expect, seq)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x, ts, pad, expect, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
