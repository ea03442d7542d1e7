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
tile's compute time, so it is issued once, un-bucketed.
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


def gather_tiles(results, classes, batch, channel, scale, group=None):
    """All-gather every rank's upscaled tiles with ONE collective.

    results: {(h,w): tensor (n_owned*batch, channel, h*s, w*s)} of THIS rank, classes in `classes` order.
    Returns a list (one entry per rank) of dicts with the same structure.
    Each rank's payload is flattened into one buffer, padded to the largest rank payload (sizes follow
    from the deterministic partition, so no size exchange is needed)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = []
    for r in range(world):
        owned = tiling.partition(classes, r, world)
        sizes.append(sum(_class_numel(hw, len(tl), batch, channel, scale) for hw, tl in owned.items()))
    cap = max(sizes)
    ref = next(iter(results.values()))
    flat = torch.zeros(cap, dtype=ref.dtype, device=ref.device)
    off = 0
    for hw in classes:
        t = results[hw].reshape(-1)
        flat[off:off + t.numel()] = t
        off += t.numel()
    assert off == sizes[rank], (off, sizes[rank])
    bufs = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(bufs, flat, group=group)
    out = []
    for r in range(world):
        owned = tiling.partition(classes, r, world)
        d, off = {}, 0
        for hw, tl in owned.items():
            n = _class_numel(hw, len(tl), batch, channel, scale)
            d[hw] = bufs[r][off:off + n].reshape(len(tl) * batch, channel, hw[0] * scale, hw[1] * scale)
            off += n
        out.append(d)
    return out


def test_tile_parallel(net, x, tile_size=240, tile_pad=16, group=None):
    """`net.test_tile` sharded over the process group; every rank returns the full upscaled image."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return net.test_tile(x, tile_size, tile_pad)
    return net.test_tile(x, tile_size, tile_pad, rank=dist.get_rank(group), world_size=dist.get_world_size(group),
                         gather=lambda res, cls, b, c, s: gather_tiles(res, cls, b, c, s, group))
