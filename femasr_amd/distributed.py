"""Tile-parallel multi-GPU inference: one process per GPU, RCCL over xGMI.

New functionality relative to the reference, whose inference is single-device
(basicsr/models/femasr_model.py:229-232) and runs tiles sequentially
(femasr_arch.py:405-429).  Tiles are independent units (every op is
per-sample), so ranks compute disjoint shares of each shape class with NO
collective on the data path; the only exchange is ONE all-gather of the
upscaled tiles per image, after which every rank pastes the full canvas
exactly as `test_tile` does (overlap-discard).

xGMI is point-to-point (7 links/GPU): a single large all-gather lets RCCL use
all links at once; the payload (3 MiB fp32 per 512x512 tile) is <3 % of the
tile's compute time, so it is issued once, un-bucketed, as
`all_gather_into_tensor` on a persistent (world, cap) buffer: tiles are written
straight into this rank's send slab by the batched `test()` calls' consumers
(one copy per class), nothing is zero-filled and nothing is re-copied on receive
— the per-rank results are views into the receive buffer.
"""
import os

import torch
import torch.distributed as dist

from . import tiling


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). Returns (rank, world, local_rank)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'     # 'nccl' IS RCCL on ROCm
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _class_numel(hw, count, batch, channel, scale):
    return count * batch * channel * hw[0] * scale * hw[1] * scale


class TileGather:
    """Persistent send/receive buffers for the all-gather of one image geometry.

    Sizes follow from the deterministic partition (tiling.partition), so no size exchange is needed; every
    rank's slab is `cap` = the largest rank payload (ranks with fewer tiles leave their tail unused)."""

    def __init__(self, classes, batch, channel, scale, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.classes, self.batch, self.channel, self.scale = classes, batch, channel, scale
        self.owned = [tiling.partition(classes, r, self.world) for r in range(self.world)]
        self.sizes = [sum(_class_numel(hw, len(tl), batch, channel, scale) for hw, tl in o.items()) for o in self.owned]
        self.cap = max(max(self.sizes), 1)
        self.send = torch.empty(self.cap, dtype=dtype, device=device)
        self.recv = torch.empty((self.world, self.cap), dtype=dtype, device=device)

    def _views(self, flat, r):
        d, off = {}, 0
        for hw, tl in self.owned[r].items():
            n = _class_numel(hw, len(tl), self.batch, self.channel, self.scale)
            d[hw] = flat[off:off + n].view(len(tl) * self.batch, self.channel, hw[0] * self.scale, hw[1] * self.scale)
            off += n
        return d

    def send_views(self):
        """{(h,w): view into this rank's send slab}: batched `test()` results are copied (or written) here."""
        return self._views(self.send, self.rank)

    def gather(self, async_op=False):
        """ONE collective.  Returns (list of per-rank {(h,w): view into the receive buffer}, work handle or None)."""
        work = dist.all_gather_into_tensor(self.recv.view(-1), self.send, group=self.group, async_op=async_op)
        return [self._views(self.recv[r], r) for r in range(self.world)], work


_gathers = {}


def gather_tiles(results, classes, batch, channel, scale, group=None):
    """All-gather every rank's upscaled tiles with ONE `all_gather_into_tensor` on persistent buffers.

    results: {(h,w): tensor (n_owned*batch, channel, h*s, w*s)} of THIS rank.
    Returns a list (one entry per rank) of dicts with the same structure (views into the receive buffer,
    valid until the next call with the same geometry)."""
    ref = next(iter(results.values()))
    key = (tuple((hw, tuple(t.index for t in tl)) for hw, tl in classes.items()), batch, channel, scale, ref.dtype,
           str(ref.device), id(group))
    tg = _gathers.get(key)
    if tg is None:
        _gathers.clear()            # one geometry at a time keeps the footprint bounded
        tg = _gathers[key] = TileGather(classes, batch, channel, scale, ref.dtype, ref.device, group)
    for hw, dst in tg.send_views().items():
        dst.copy_(results[hw])
    out, _ = tg.gather()
    return out


def test_tile_parallel(net, x, tile_size=240, tile_pad=16, group=None):
    """`net.test_tile` sharded over the process group; every rank returns the full upscaled image."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return net.test_tile(x, tile_size, tile_pad)
    return net.test_tile(x, tile_size, tile_pad, rank=dist.get_rank(group), world_size=dist.get_world_size(group),
                         gather=lambda res, cls, b, c, s: gather_tiles(res, cls, b, c, s, group))
