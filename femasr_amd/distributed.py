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
`all_gather_into_tensor` on a persistent (world, cap) buffer.  The batched
`test()` calls write their tiles STRAIGHT into this rank's send slab
(`FeMaSRNet.test(x, out=slab_view)`: the forward's last kernel stores there -
`TileExchange.send_views`), nothing is zero-filled and nothing is re-copied on
receive - the per-rank results are views into the receive buffer.  A rank that
does not need the image (`paste=False`) joins the collective and skips the canvas.

Not measured: no node with more than one GPU was available to this build; the
scaling curve is the driver's to record (bench.py --gpus N).
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

    def __init__(self, classes, batch, channel, scale, dtype, device, group=None, layout='nchw'):
        self.group = group
        self.layout = layout            # 'nchw': fp32 tiles (n, C, H, W) of test_tile; 'nhwc': uint8 tiles (n, H, W, C) of test_tile_u8
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.classes, self.batch, self.channel, self.scale = classes, batch, channel, scale
        self.owned = tiling.assign(classes, self.world, scale)
        self.sizes = [sum(_class_numel(hw, len(tl), batch, channel, scale) for hw, tl in o.items()) for o in self.owned]
        self.cap = max(max(self.sizes), 1)
        self.send = torch.empty(self.cap, dtype=dtype, device=device)
        self.recv = torch.empty((self.world, self.cap), dtype=dtype, device=device)

    def _views(self, flat, r):
        d, off = {}, 0
        for hw, tl in self.owned[r].items():
            n = _class_numel(hw, len(tl), self.batch, self.channel, self.scale)
            shp = (len(tl) * self.batch, self.channel, hw[0] * self.scale, hw[1] * self.scale) if self.layout == 'nchw' else \
                  (len(tl) * self.batch, hw[0] * self.scale, hw[1] * self.scale, self.channel)
            d[hw] = flat[off:off + n].view(shp)
            off += n
        return d

    def send_views(self):
        """{(h,w): view into this rank's send slab}: batched `test()` results are copied (or written) here."""
        return self._views(self.send, self.rank)

    def gather(self, async_op=False):
        """ONE collective.  Returns (list of per-rank {(h,w): view into the receive buffer}, work handle or None)."""
        work = dist.all_gather_into_tensor(self.recv.view(-1), self.send, group=self.group, async_op=async_op)
        return [self._views(self.recv[r], r) for r in range(self.world)], work


class StepGather:
    """Weak-scaling serving loop (bench.py tiles16: every rank upscales its own batch each step): the all-gather of step k runs on
    the collective's stream while step k+1 computes.  Double-buffered persistent send and receive buffers; `send(k)` is the tensor
    the forward of step k writes into (`net.test(x, out=...)`), `launch(k)` starts the collective of step k and - before the
    buffer pair of step k-1 can be re-used by step k+1 - waits for that older one; `result(k)` = (world, *shape) view, valid
    after `wait_all()` or once `launch(k+1)` has returned."""

    def __init__(self, shape, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self._send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        self._recv = [torch.empty((self.world,) + tuple(shape), dtype=dtype, device=device) for _ in range(2)]
        self._pending = []

    def send(self, k):
        return self._send[k & 1]

    def launch(self, k):
        w = dist.all_gather_into_tensor(self._recv[k & 1].view(-1), self._send[k & 1].view(-1), group=self.group, async_op=True)
        self._pending.append(w)
        if len(self._pending) > 1:
            self._pending.pop(0).wait()

    def result(self, k):
        return self._recv[k & 1]

    def wait_all(self):
        while self._pending:
            self._pending.pop(0).wait()


class TileExchange:
    """The `gather` object `FeMaSRNet.test_tile` takes: owns the persistent buffers of ONE image geometry at a time.

    `send_views(...)` hands test_tile this rank's send slab as {(h, w): tensor}, so the tiles are produced in place;
    calling the object runs the collective (tiles handed over in other tensors - a stand-in network without `out=` - are copied
    into the slab first)."""

    def __init__(self, group=None):
        self.group = group
        self._key = None
        self._tg = None

    def _get(self, classes, batch, channel, scale, dtype, device, layout='nchw'):
        # (world size and rank are part of the key: after destroy_process_group + a new init in the same process - or a freed group's
        # id() coming back - a TileGather built for another world would hand stale sizes to the collective)
        key = (tuple((hw, tuple(t.index for t in tl)) for hw, tl in classes.items()), batch, channel, scale, dtype, str(device),
               dist.get_world_size(self.group), dist.get_rank(self.group), layout)
        if key != self._key:
            self._tg = None                 # one geometry at a time keeps the footprint bounded
            self._tg = TileGather(classes, batch, channel, scale, dtype, device, self.group, layout)
            self._key = key
        return self._tg

    def send_views(self, classes, batch, channel, scale, dtype, device, layout='nchw'):
        return self._get(classes, batch, channel, scale, dtype, torch.device(device), layout).send_views()

    def __call__(self, results, classes, batch, channel, scale, layout='nchw'):
        ref = next(iter(results.values()))
        tg = self._get(classes, batch, channel, scale, ref.dtype, ref.device, layout)
        for hw, dst in tg.send_views().items():
            src = results[hw]
            if src.data_ptr() != dst.data_ptr() or src.numel() != dst.numel():
                dst.copy_(src)
        out, _ = tg.gather()
        return out


_exchanges = {}


def _exchange_for(group):
    """The cached TileExchange of a process group; entries of groups that no longer exist (or whose id() was re-used for a group of
    another size / rank) are dropped."""
    sig = (id(group), dist.get_world_size(group), dist.get_rank(group))
    for k in [k for k in _exchanges if k[0] == sig[0] and k != sig]:
        del _exchanges[k]
    if sig not in _exchanges:               # (not setdefault(sig, TileExchange(group)): its argument is built on every call)
        _exchanges[sig] = TileExchange(group)
    return _exchanges[sig]


def gather_tiles(results, classes, batch, channel, scale, group=None, layout='nchw'):
    """All-gather every rank's upscaled tiles with ONE `all_gather_into_tensor` on persistent buffers (functional form of
    TileExchange; the tiles are copied into the send slab unless they already live there).

    results: {(h,w): tensor (n_owned*batch, channel, h*s, w*s)} of THIS rank.
    Returns a list (one entry per rank) of dicts with the same structure (views into the receive buffer,
    valid until the next call with the same geometry)."""
    return _exchange_for(group)(results, classes, batch, channel, scale, layout)


def test_tile_parallel(net, x, tile_size=240, tile_pad=16, group=None, root_only=False):
    """`net.test_tile` sharded over the process group.  Every rank returns the full upscaled image, or - root_only=True - only
    rank 0 pastes it (the others return None): one canvas write per job instead of one per rank.
    A uint8 image ((H,W,3) / (B,H,W,3), the CLI's data type) takes the uint8 path (`net.test_tile_u8`): tiles are produced, gathered
    and pasted as bytes - a quarter of the fp32 path's xGMI payload and canvas traffic."""
    fn = net.test_tile_u8 if x.dtype == torch.uint8 else net.test_tile
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return fn(x, tile_size, tile_pad)
    ex = _exchange_for(group)
    rank = dist.get_rank(group)
    return fn(x, tile_size, tile_pad, rank=rank, world_size=dist.get_world_size(group), gather=ex,
              paste=(rank == 0 or not root_only))
