"""Tile geometry of the reference's tiled inference and its rank partition.

Pure integer host logic (no torch, no GPU) so it is unit-testable on CPU.

Reference behaviour restated (basicsr/archs/femasr_arch.py:387-447
`FeMaSRNet.test_tile`): tiles are enumerated row-major over
ceil(H/ts) x ceil(W/ts); each tile's input window is the tile extended by
`tile_pad` pixels and clamped to the image; each window goes through `test()`
on its own; the UN-padded centre of the result is pasted (overlap-discard, no
blending) into a zero-initialised canvas.

What is new here (the reference runs tiles one at a time on one device):
tiles are grouped by input-window shape ("shape class") so each class runs as
ONE batched `test()` call — legal because every op in the network is
per-sample (GroupNorm/LayerNorm/attention), verified bit-identical in
SURVEY 8c — and the tiles are spread over the ranks by work (padded pixels), longest-processing-time first (`assign`).
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class Tile:
    index: int          # row-major tile number (y * tiles_x + x)
    # input window (with halo, clamped)            [y0p:y1p, x0p:x1p]
    y0p: int
    y1p: int
    x0p: int
    x1p: int
    # tile body in image coordinates               [y0:y1, x0:x1]
    y0: int
    y1: int
    x0: int
    x1: int

    @property
    def in_hw(self) -> Tuple[int, int]:
        return (self.y1p - self.y0p, self.x1p - self.x0p)

    def out_src(self, s: int):
        """(ys, ye, xs, xe) of the body inside the upscaled window."""
        ys, xs = (self.y0 - self.y0p) * s, (self.x0 - self.x0p) * s
        return ys, ys + (self.y1 - self.y0) * s, xs, xs + (self.x1 - self.x0) * s

    def out_dst(self, s: int):
        """(ys, ye, xs, xe) of the body on the upscaled canvas."""
        return self.y0 * s, self.y1 * s, self.x0 * s, self.x1 * s


def enumerate_tiles(height: int, width: int, tile_size: int, tile_pad: int) -> List[Tile]:
    tiles = []
    ny, nx = math.ceil(height / tile_size), math.ceil(width / tile_size)
    for ty in range(ny):
        for tx in range(nx):
            x0, y0 = tx * tile_size, ty * tile_size
            x1, y1 = min(x0 + tile_size, width), min(y0 + tile_size, height)
            tiles.append(Tile(ty * nx + tx,
                              max(y0 - tile_pad, 0), min(y1 + tile_pad, height),
                              max(x0 - tile_pad, 0), min(x1 + tile_pad, width),
                              y0, y1, x0, x1))
    return tiles


def shape_classes(tiles: List[Tile]) -> "OrderedDict[Tuple[int, int], List[Tile]]":
    """Group by input-window (h, w), classes ordered by first appearance, tiles in row-major order."""
    classes: "OrderedDict[Tuple[int, int], List[Tile]]" = OrderedDict()
    for t in tiles:
        classes.setdefault(t.in_hw, []).append(t)
    return classes


def rank_slice(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of n items for `rank`; the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


CALL_OVERHEAD_PX = 3 * 144 * 144 // 4      # fixed cost of one more batched test() call, in padded pixels (~0.75 of a 128x128 tile)


def tile_cost(hw: Tuple[int, int], scale: int = 4) -> int:
    """Work of one `test()` call on a window of this shape: the pixels it runs on after test()'s mirror pad."""
    hp, wp = padded_hw(hw[0], hw[1], scale)
    return hp * wp


def assign(classes: Dict[Tuple[int, int], List[Tile]], world: int, scale: int = 4) -> List["OrderedDict[Tuple[int, int], List[Tile]]"]:
    """Balanced ownership of all ranks: classes in order of descending per-tile cost (ties: first appearance), every tile to the
    rank with the least work so far (ties: the lowest rank) - the longest-processing-time rule over (class, padded pixels).
    A rank's tiles of one class are consecutive in the class's row-major list, in rank order, so they still run as batched calls
    and paste in the reference's order.  Deterministic: every rank computes the same table.  (Round 4 split every class on its
    own and gave each remainder to the lowest ranks: 15/10/../6 tiles at 240/16 on 8 ranks, a 5.9x bound before a byte moves.)"""
    load = [0] * world
    counts = {hw: [0] * world for hw in classes}
    order = sorted(classes, key=lambda hw: -tile_cost(hw, scale))          # (stable: equal costs keep first-appearance order)
    for hw in order:
        c = tile_cost(hw, scale)
        for _ in classes[hw]:
            # a rank's FIRST tile of a class opens one more batched call: CALL_OVERHEAD_PX of latency-bound launches (measured: a
            # 128x128 tile takes 7.7 ms alone and 4.5 ms inside a batch of 16) - so a class is not scattered needlessly
            r = min(range(world), key=lambda k: (load[k] + c + (0 if counts[hw][k] else CALL_OVERHEAD_PX), k))
            load[r] += c + (0 if counts[hw][r] else CALL_OVERHEAD_PX)
            counts[hw][r] += 1
    out = [OrderedDict() for _ in range(world)]
    for hw, tl in classes.items():                                        # class order = first appearance (the reference's order)
        lo = 0
        for r in range(world):
            out[r][hw] = tl[lo:lo + counts[hw][r]]
            lo += counts[hw][r]
    return out


def partition(classes: Dict[Tuple[int, int], List[Tile]], rank: int, world: int, scale: int = 4):
    """Per-class tile lists owned by `rank` (see `assign`)."""
    return assign(classes, world, scale)[rank]


def balance_bound(classes: Dict[Tuple[int, int], List[Tile]], world: int, scale: int = 4) -> float:
    """Total work / the busiest rank's work: the speed-up the partition allows at `world` ranks before communication."""
    own = assign(classes, world, scale)
    per = [sum(tile_cost(hw, scale) * len(tl) for hw, tl in o.items()) for o in own]
    return sum(per) / max(max(per), 1)


def padded_hw(h: int, w: int, scale: int) -> Tuple[int, int]:
    """`test()` geometry (femasr_arch.py:454-458): ALWAYS pads to the next multiple of wsz, even if divisible."""
    wsz = 8 // scale * 8
    return (h // wsz + 1) * wsz, (w // wsz + 1) * wsz
