"""The N>1 path on CPU: world_size-2 (and 3, uneven split) gloo processes shard one frame's rays,
render their range and all-gather the tiles; the result must equal the single-process render.
The per-range renderer here is the CPU oracle (test infrastructure) -- the sharding/gather code
under test is exactly what bench.py and the GPU path use."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from synthetic import synth
from diner_amd.dist import all_gather_tiles, render_frame_sharded, shard_bounds


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 64, 262144, 327680):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene_and_oracle():
    from oracle.oracle import Oracle
    sc = synth.make_scene(16, 16, 2, seed=0, feature_padding=4)
    w = synth.make_mlp_weights(1)
    return sc, Oracle(sc, w, threads=1)


K, NC, G = 8, 64, 3


def _render_range(orc, sc, rays_np, noise, lo):
    n = rays_np.shape[1]
    sub = tuple(x[lo:lo + n] for x in noise)
    out = orc.render(rays_np, NC, K, G, sub, white_bkgd=sc.white_bkgd)
    return torch.from_numpy(np.concatenate([out["rgb"], out["depth"][:, None]], 1))


def _worker(rank, world, port, n_rays, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, orc = _scene_and_oracle()
        rays = torch.from_numpy(sc.target_rays()[:, :n_rays])
        noise = synth.make_noise(n_rays, NC, G, K, seed=3)
        lo, _ = shard_bounds(n_rays, world, rank)
        frame = render_frame_sharded(lambda r: _render_range(orc, sc, r.numpy(), noise, lo), rays, world, rank)
        q.put((rank, frame.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rays", [(2, 64), (3, 50)])
def test_sharded_frame_equals_single_process(world, n_rays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc, orc = _scene_and_oracle()
    rays = sc.target_rays()[:, :n_rays]
    noise = synth.make_noise(n_rays, NC, G, K, seed=3)
    want = _render_range(orc, sc, rays, noise, 0).numpy()
    for r in range(world):
        np.testing.assert_array_equal(got[r], want)


def _worker_overlapped(rank, world, port, n_rays, q):
    """three frames (different noise) through the double-buffered gather: frame i's all-gather is in flight while i+1 renders"""
    from diner_amd.dist import OverlappedGather
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, orc = _scene_and_oracle()
        rays = sc.target_rays()[:, :n_rays]
        lo, hi = shard_bounds(n_rays, world, rank)
        og = OverlappedGather(n_rays, world, rank, 4, "cpu")
        frames = []
        for f in range(3):
            noise = synth.make_noise(n_rays, NC, G, K, seed=10 + f)
            og.tile().copy_(_render_range(orc, sc, rays[:, lo:hi], noise, lo))
            prev = og.submit()
            assert (prev is None) == (f == 0)
            if prev is not None:
                frames.append(prev.clone())
        frames.append(og.flush().clone())
        assert og.flush() is None
        q.put((rank, np.stack([f.numpy() for f in frames])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rays", [(2, 64), (3, 50)])
def test_overlapped_gather_frames_equal_single_process(world, n_rays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlapped, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc, orc = _scene_and_oracle()
    rays = sc.target_rays()[:, :n_rays]
    want = np.stack([_render_range(orc, sc, rays, synth.make_noise(n_rays, NC, G, K, seed=10 + f), 0).numpy() for f in range(3)])
    for r in range(world):
        np.testing.assert_array_equal(got[r], want)
