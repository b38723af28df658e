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


# ---------------------------------------------------------------------------------------------------------------
# volume sharding of the K-volume boost path: all-to-all of (raw, z, mask), sharded blend, tile all-gather
# ---------------------------------------------------------------------------------------------------------------
def _volume_worker(rank, world, port, K, n_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import enerf as O   # stand-in renderer / blend for the CPU test
        torch.manual_seed(0)            # the same "scene" on every rank
        Ns = 2
        raws = torch.rand(1, K, n_rays, Ns, 4)
        zs = torch.rand(1, K, n_rays, Ns) + 2
        ms = torch.randint(0, 4, (1, K, n_rays, Ns)).float() / 3
        want = O.blend(raws, O.normalise_masks(ms), zs)
        vs = sharding.VolumeShard(world, rank, K, n_rays)
        b, e = vs.ray_range
        mine = [t[:, vs.volumes, b:e].contiguous() for t in (raws, zs, ms)]          # what this rank "rendered"
        r2, z2, m2 = vs.exchange(*mine)
        sb, se = vs.sub_range()
        ok_x = (torch.equal(r2, raws[:, :, sb:se]) and torch.equal(z2, zs[:, :, sb:se]) and torch.equal(m2, ms[:, :, sb:se]))
        part = O.blend(r2, O.normalise_masks(m2), z2)
        frame = vs.gather_tiles(part["rgb"], part["depth"])
        ok_f = torch.equal(frame[:, :3], want["rgb"][0]) and torch.equal(frame[:, 3], want["depth"][0])
        q.put((rank, bool(ok_x), bool(ok_f), tuple(frame.shape), vs.volumes, (b, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K,n_rays", [(2, 4, 101), (4, 2, 203)])
def test_volume_sharding_exchange_and_blend(world, K, n_rays):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_volume_worker, args=(r, world, port, K, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = set()
    for rank, ok_x, ok_f, shape, vols, rr in res:
        assert ok_x and ok_f, (rank, ok_x, ok_f)
        assert shape == (n_rays, 4)
        seen.update((k, rr) for k in vols)
    G = min(world, K)
    assert len(seen) == K * (world // G)          # every (volume, ray group) pair has exactly one owner


# ---------------------------------------------------------------------------------------------------------------
# bench.py's OWN exchange wiring (bench.make_exchange: the function its timed step calls), with a stub renderer
# ---------------------------------------------------------------------------------------------------------------
class _StubNet:
    """What bench.make_exchange touches of a network: the ray / volume shard attributes and the K-volume fusion."""
    ray_range = None
    volume_ids = None

    @staticmethod
    def merge_mlp_outputs(raws, ms, zs):
        from oracle import enerf as O
        return O.blend(raws, O.normalise_masks(ms), zs)


def _bench_worker(rank, world, port, n_rays, K, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench                      # importing bench.py needs no GPU
        from oracle import enerf as O
        torch.manual_seed(0)              # the same "scene" on every rank
        Ns, level = 2, 1
        raws = torch.rand(1, K, n_rays, Ns, 4)
        zs = torch.rand(1, K, n_rays, Ns) + 2
        ms = torch.randint(0, 4, (1, K, n_rays, Ns)).float() / 3
        full = O.blend(raws, O.normalise_masks(ms), zs)
        ok = {}
        # --shard rays: the network renders [begin, end) of the frame, finish() reassembles it
        net = _StubNet()
        finish, gather, vshard = bench.make_exchange("rays", world, rank, n_rays, "cpu", net, {"net": "enerf"}, 1, level)
        b, e = net.ray_range
        out = {"rgb_level1": full["rgb"][:, b:e], "depth_level1": full["depth"][:, b:e]}
        frame = finish(out)
        ok["rays"] = vshard is None and torch.equal(frame[:, :3], full["rgb"][0]) and torch.equal(frame[:, 3], full["depth"][0])
        # --shard views (synchronous and pipelined): every rank its own frame, all frames everywhere
        for pipelined in (False, True):
            net = _StubNet()
            finish, gather, _ = bench.make_exchange("views", world, rank, n_rays, "cpu", net, {"net": "enerf"}, 1, level,
                                                    pipelined=pipelined)
            assert net.ray_range is None
            got = None
            for i in range(3):
                got = finish({"rgb_level1": full["rgb"] * (rank + 1 + i), "depth_level1": full["depth"] + rank})
            if pipelined:
                got = gather.flush()
            ok[f"views{int(pipelined)}"] = all(torch.equal(got[r, :, :3], full["rgb"][0] * (r + 3)) and
                                                 torch.equal(got[r, :, 3], full["depth"][0] + r) for r in range(world))
        # --shard volumes: volume subsets x ray ranges, all-to-all, sharded fusion, tile gather
        net = _StubNet()
        finish, gather, vshard = bench.make_exchange("volumes", world, rank, n_rays, "cpu", net, {"net": "boost_enerf"}, K, level)
        b, e = net.ray_range
        stacks = tuple(t[:, net.volume_ids, b:e].contiguous() for t in (raws, zs, ms))
        frame = finish({"stacks_level1": stacks})
        ok["volumes"] = gather is None and torch.equal(frame[:, :3], full["rgb"][0]) and torch.equal(frame[:, 3], full["depth"][0])
        # world-1 semantics of the same function: identity
        f1, g1, v1 = bench.make_exchange("views", 1, 0, n_rays, "cpu", _StubNet(), {"net": "enerf"}, 1, level)
        ok["identity"] = g1 is None and v1 is None and f1(out) is out
        q.put((rank, {k: bool(v) for k, v in ok.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K,n_rays", [(2, 4, 101), (4, 2, 96)])
def test_bench_exchange_wiring(world, K, n_rays):
    """bench.py --gpus N --shard {rays, views, volumes}: the exchange its timed step runs, on gloo."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, n_rays, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok.values()), (rank, ok)


# ---------------------------------------------------------------------------------------------------------------
# fine-tune leg under DDP (lib/train/trainers/trainer.py:15-22: SyncBatchNorm conversion + DistributedDataParallel
# with find_unused_parameters around the loss wrapper): construction + one step on 2 ranks.  The HIP forward cannot
# run on CPU, so the network's forward is replaced by a surrogate that touches every parameter; what is under test
# is the wrapping of OUR module tree (state-dict keys survive the conversion, gradients are averaged, the ranks'
# parameters stay identical after train_step).
# ---------------------------------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from boostmvsnerfs_amd.config import make_cfg, set_cfg
        set_cfg(make_cfg("enerf_pretrain"))
        from boostmvsnerfs_amd.networks.enerf.network import Network
        from boostmvsnerfs_amd.train import NetworkWrapper, make_optimizer, train_step
        torch.manual_seed(0)
        net = Network()
        keys = list(net.state_dict().keys())

        def surrogate(batch):      # every parameter contributes; rank-dependent data
            s = sum((p * p).mean() for p in net.parameters())
            n = batch["rgb_1"].shape[1]
            return {"rgb_level0": (batch["rgb_0"] * 0 + s), "rgb_level1": batch["rgb_1"] * 0 + s * (1.0 + batch["scale"])}
        net.forward = surrogate
        wrapper = NetworkWrapper(net)
        # trainer.py:17: the conversion must keep every state-dict key (checkpoints stay loadable); torch only lets
        # SyncBatchNorm modules RUN under DDP on GPU tensors, so the CPU step below keeps the plain batch norms
        # (tests/test_gpu_training.py runs the converted wrapper under DDP on the GPU)
        import copy
        conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(wrapper))
        n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in conv.modules())
        same_keys = [k.replace("net.", "", 1) for k in conv.state_dict().keys()] == keys
        ddp = torch.nn.parallel.DistributedDataParallel(wrapper, find_unused_parameters=True)
        same_keys = same_keys and [k.replace("module.net.", "") for k in ddp.state_dict().keys()] == keys
        opt = make_optimizer(net)
        g = torch.Generator().manual_seed(rank)
        batch = {"rgb_0": torch.rand(1, 16, 3, generator=g), "rgb_1": torch.rand(1, 64, 3, generator=g), "scale": float(rank)}
        loss, stats = train_step(ddp, opt, batch)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        in_sync = all(torch.equal(gathered[0], t) for t in gathered)
        moved = bool((flat != torch.cat([p.reshape(-1) for p in Network().parameters()])).any()) if rank == 0 else True
        q.put((rank, same_keys, n_sync, in_sync, bool(torch.isfinite(loss)), moved))
    finally:
        dist.destroy_process_group()


def test_finetune_wrapper_under_ddp_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_keys, n_sync, in_sync, finite, moved in res:
        assert same_keys, "SyncBatchNorm conversion / DDP changed the state-dict keys"
        assert n_sync > 0 and in_sync and finite and moved, (rank, n_sync, in_sync, finite, moved)
