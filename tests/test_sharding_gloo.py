"""N>1 path on CPU: world_size-2 gloo processes shard rays / views and all-gather
the rendered tiles exactly as bench.py does over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from boostmvsnerfs_amd import sharding


def test_ray_slices_partition():
    for n in (0, 1, 7, 327680, 327681):
        for world in (1, 2, 3, 8):
            sl = sharding.all_slices(n, world)
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            sizes = [e - b for b, e in sl]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.ray_slice(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import enerf as O   # stand-in renderer for the CPU test
        torch.manual_seed(0)            # same "scene" on every rank
        raw = torch.rand(1, n_rays, 2, 4)
        z = torch.rand(1, n_rays, 2) + 2
        full = O.composite(raw, z)
        tg = sharding.TileGather(world)
        # rays sharding: each rank composites its slice, all-gather reassembles the frame
        b, e = sharding.ray_slice(n_rays, world, rank)
        part = O.composite(raw[:, b:e], z[:, b:e])
        frame = tg.all_gather_ray_tiles(part["rgb"], part["depth"], n_rays)
        ok_rays = torch.equal(frame[:, :3], full["rgb"][0]) and torch.equal(frame[:, 3], full["depth"][0])
        # views sharding: rank r renders "its" frame (scaled copy), everyone receives all frames
        mine = {"rgb": full["rgb"] * (rank + 1), "depth": full["depth"] + rank}
        frames = sharding.TileGather(world).all_gather_frames(mine["rgb"], mine["depth"])
        ok_views = all(torch.equal(frames[r, :, :3], full["rgb"][0] * (r + 1)) and
                       torch.equal(frames[r, :, 3], full["depth"][0] + r) for r in range(world))
        # pipelined views sharding (bench.py default): call i returns the frames of call i-1, flush() the last
        tp = sharding.TileGather(world)
        got = []
        for i in range(4):
            g = tp.all_gather_frames_pipelined(full["rgb"] * (rank + 1 + 10 * i), full["depth"] + rank + i)
            got.append(None if g is None else g.clone())      # the two tile buffers are reused every other call
        assert got[0] is None
        got = got[1:] + [tp.flush()]
        ok_pipe = got[0] is not None and all(
            torch.equal(got[i][r, :, :3], full["rgb"][0] * (r + 1 + 10 * i)) and
            torch.equal(got[i][r, :, 3], full["depth"][0] + r + i) for i in range(4) for r in range(world))
        ok_views = ok_views and ok_pipe and sharding.TileGather(world).flush() is None
        q.put((rank, bool(ok_rays), bool(ok_views), tuple(frame.shape), tuple(frames.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rays", [96, 101])
def test_two_rank_gather(n_rays):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_rays, ok_views, fshape, vshape in res:
        assert ok_rays and ok_views, (rank, ok_rays, ok_views)
        assert fshape == (n_rays, 4) and vshape == (world, n_rays, 4)
