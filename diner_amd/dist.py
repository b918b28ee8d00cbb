"""Multi-GPU sharding of the render path: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Rays are independent and cost the same (always NC candidates and K MLP evaluations, no early
termination), so a frame is split into ``world`` contiguous, equally sized ray ranges; maps,
latent and weights are replicated.  The only exchange step of the path is one all-gather of the
rendered ``[rays, 4]`` tiles (rgb + depth; 4.2 MB per frame at 512x512) -- latency-bound on xGMI,
so it is issued as ONE collective per frame rather than per chunk.  The reference has no
counterpart (its render path never leaves one GPU); this is the §8(e) design.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split of ``n`` rays: the first ``n % world`` ranks get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_tiles(tile: torch.Tensor, n_total: int, world: int, group=None) -> torch.Tensor:
    """Gather per-rank ``[n_rank, C]`` tiles (ranges from :func:`shard_bounds`) into ``[n_total, C]``
    on every rank with a single collective (tiles are padded to the largest range)."""
    if world == 1:
        return tile
    q, r = divmod(n_total, world)
    n_max = q + (1 if r else 0)
    padded = tile
    if tile.shape[0] != n_max:
        padded = torch.zeros((n_max, tile.shape[1]), dtype=tile.dtype, device=tile.device)
        padded[:tile.shape[0]] = tile
    out = torch.empty((world * n_max, tile.shape[1]), dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if r == 0:
        return out
    parts = []
    for k in range(world):
        lo, hi = shard_bounds(n_total, world, k)
        parts.append(out[k * n_max:k * n_max + (hi - lo)])
    return torch.cat(parts, 0)


class OverlappedGather:
    """The §8(e) overlap: double-buffered tiles, so that the all-gather of frame i travels (RCCL's own stream) while frame i+1
    renders.  ``tile()`` is the buffer to render the current frame's ``[n_rank, C]`` rows into; ``submit()`` starts its
    asynchronous all-gather and returns the PREVIOUS frame, complete, as ``[n_total, C]`` (None for the first);
    ``flush()`` returns the last one.  Uneven ranges are padded inside the buffers (no copy per frame)."""

    def __init__(self, n_total: int, world: int, rank: int, channels: int, device, dtype=torch.float32, group=None):
        self.n_total, self.world, self.group = n_total, world, group
        lo, hi = shard_bounds(n_total, world, rank)
        self.n_mine = hi - lo
        q, r = divmod(n_total, world)
        self.n_max = q + (1 if r else 0)
        self._tiles = [torch.zeros((self.n_max, channels), dtype=dtype, device=device) for _ in range(2)]
        self._outs = [torch.empty((world * self.n_max, channels), dtype=dtype, device=device) for _ in range(2)]
        self._work = [None, None]
        self._cur = 0
        self._inflight = None          # buffer index of the frame whose gather is travelling

    def tile(self) -> torch.Tensor:
        return self._tiles[self._cur][:self.n_mine]

    def _finish(self, b):
        if self._work[b] is not None:
            self._work[b].wait()
            self._work[b] = None
        if self.world == 1:
            return self._tiles[b][:self.n_mine]
        out = self._outs[b]
        if self.n_total == self.world * self.n_max:
            return out
        parts = []
        for k in range(self.world):
            lo, hi = shard_bounds(self.n_total, self.world, k)
            parts.append(out[k * self.n_max:k * self.n_max + (hi - lo)])
        return torch.cat(parts, 0)

    def submit(self):
        b = self._cur
        if self.world > 1:
            self._work[b] = dist.all_gather_into_tensor(self._outs[b], self._tiles[b], group=self.group, async_op=True)
        prev = None if self._inflight is None else self._finish(self._inflight)
        self._inflight, self._cur = b, 1 - b
        return prev

    def flush(self):
        if self._inflight is None:
            return None
        frame, self._inflight = self._finish(self._inflight), None
        return frame


def render_frame_sharded(render_fn: Callable[[torch.Tensor], torch.Tensor], rays: torch.Tensor,
                         world: int, rank: int, group=None) -> torch.Tensor:
    """Strong-scaling frame render: this rank renders its contiguous range of ``rays`` [1,NR,8]
    with ``render_fn(rays_range) -> [n,4]`` (rgb, depth) and every rank receives the whole
    ``[NR,4]`` frame."""
    NR = rays.shape[1]
    lo, hi = shard_bounds(NR, world, rank)
    tile = render_fn(rays[:, lo:hi])
    return all_gather_tiles(tile, NR, world, group=group)
