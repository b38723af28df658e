"""Multi-GPU sharding of the render path (one process per GPU, torch.distributed;
backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Two ways the path shards, neither needs a collective
inside the kernels:
  * views: every rank renders a different target frame; the only exchange is
    an all-gather of the finished (rgb, depth) tiles (N*4 floats per rank).
  * rays:  one frame, contiguous row-major ray ranges per rank (the front end --
    features, sweep, regulariser -- is replicated: no halo logic, no traffic),
    then the same all-gather reassembles the frame.
  * volumes (the K-volume boost networks): the K cost volumes of a frame are independent until the fusion
    (boost_enerf/network.py:172-237), so rank (g, r) builds and renders volumes {k : k % G == g} for ray slice r
    (G = min(world, K) volume groups x R = world / G ray groups; the 2-D feature net is replicated), the G ranks of a
    ray group swap (raw, z, mask) with ONE all-to-all so that each ends up with ALL K volumes for 1/G of the slice,
    fuses it (the blend kernel), and the usual tile all-gather reassembles the frame.  Unlike ray sharding this
    divides the 3-D regularisers -- 3/4 of a boost frame -- by G instead of replicating them.
Tiles are packed (rgb, depth) into one buffer so a step issues ONE collective;
at 512x640 it moves 5.2 MB per rank, latency-bound on xGMI.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def ray_slice(n_rays, world, rank):
    """Contiguous, near-equal split of [0, n_rays): rank r gets [begin, end)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n_rays, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def all_slices(n_rays, world):
    return [ray_slice(n_rays, world, r) for r in range(world)]


class TileGather:
    """Reusable buffers + one all_gather per step."""

    def __init__(self, world, n_rays=None, device="cpu", group=None):
        self.world = world
        self.group = group
        self.device = device
        self._send = None
        self._recv = None

    def _buffers(self, rows, device):
        if self._send is None or self._send.shape[0] != rows:
            self._send = torch.empty(rows, 4, device=device, dtype=torch.float32)
            self._recv = torch.empty(self.world * rows, 4, device=device, dtype=torch.float32)
        return self._send, self._recv

    def all_gather_frames(self, rgb, depth):
        """views sharding: rgb (1,N,3), depth (1,N) of this rank's frame ->
        (world, N, 4) tensor [r, g, b, depth] of every rank's frame."""
        n = rgb.shape[-2]
        send, recv = self._buffers(n, rgb.device)
        send[:, :3] = rgb.reshape(n, 3)
        send[:, 3] = depth.reshape(n)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv.view(self.world, n, 4)

    # ---- pipelined variant: the exchange of frame i overlaps the rendering of frame i+1 --------------------
    def all_gather_frames_pipelined(self, rgb, depth):
        """Same exchange, issued asynchronously on the collective's own stream with double-buffered tiles.
        Returns the frames gathered by the PREVIOUS call (None on the first): a renderer streams frames, so the
        all-gather of frame i rides under the kernels of frame i+1 instead of stalling the compute stream.
        `flush()` returns the last gathered frames."""
        n = rgb.shape[-2]
        if getattr(self, "_pipe", None) is None or self._pipe[0][0].shape[0] != n:
            dev = rgb.device
            self._pipe = [(torch.empty(n, 4, device=dev), torch.empty(self.world * n, 4, device=dev)) for _ in range(2)]
            self._work = [None, None]
            self._step = 0
        slot = self._step & 1
        self._step += 1
        if self._work[slot] is not None:       # this slot's previous exchange (two calls ago) must be done
            self._work[slot].wait()
        send, recv = self._pipe[slot]
        send[:, :3] = rgb.reshape(n, 3)
        send[:, 3] = depth.reshape(n)
        self._work[slot] = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        prev = slot ^ 1
        if self._work[prev] is None:
            return None
        self._work[prev].wait()
        return self._pipe[prev][1].view(self.world, n, 4)

    def flush(self):
        """Wait for every exchange in flight; returns the most recent gathered frames (or None)."""
        if getattr(self, "_pipe", None) is None or self._step == 0:
            return None
        for w in self._work:
            if w is not None:
                w.wait()
        last = (self._step - 1) & 1
        n = self._pipe[last][0].shape[0]
        return self._pipe[last][1].view(self.world, n, 4)

    def all_gather_ray_tiles(self, rgb, depth, n_rays):
        """rays sharding: this rank's tile (1,n_r,3)/(1,n_r) -> full frame (n_rays, 4).
        Tiles differ by at most one ray; they are padded to the largest for a single collective."""
        slices = all_slices(n_rays, self.world)
        rows = max(e - b for b, e in slices)
        n = rgb.shape[-2]
        send, recv = self._buffers(rows, rgb.device)
        send[:n, :3] = rgb.reshape(n, 3)
        send[:n, 3] = depth.reshape(n)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        parts = recv.view(self.world, rows, 4)
        if all(e - b == rows for b, e in slices):
            return parts.reshape(n_rays, 4)
        return torch.cat([parts[r, : e - b] for r, (b, e) in enumerate(slices)], 0)


class VolumeShard:
    """Cost-volume parallelism of the K-volume boost path (see the module docstring).

    rank = r * G + g; `volumes` and `ray_range` are what this rank tells its network to compute
    (`net.volume_ids`, `net.ray_range`); `exchange` turns its (1, K/G, n_r, Ns, .) stacks into the
    (1, K, n_sub, Ns, .) stacks of its ray sub-slice; `gather_tiles` reassembles the fused frame."""

    def __init__(self, world, rank, K, n_rays):
        G = min(world, K)
        if K % G or world % G:
            raise ValueError(f"volume sharding needs K ({K}) and the world size ({world}) to be multiples of min(world, K)")
        self.world, self.rank, self.K, self.G, self.R, self.n_rays = world, rank, K, G, world // G, n_rays
        self.g, self.r = rank % G, rank // G
        self.volumes = [k for k in range(K) if k % G == self.g]
        self.ray_range = ray_slice(n_rays, self.R, self.r)
        # every rank creates every group (torch.distributed requirement); mine is the one of my ray group
        self.group = None
        if self.R > 1:
            for r in range(self.R):
                grp = dist.new_group(ranks=[r * G + g for g in range(G)])
                if r == self.r:
                    self.group = grp
        n_r = self.ray_range[1] - self.ray_range[0]
        self.sub = -(-n_r // G)                                            # rays per rank after the exchange (padded)
        self.sub_max = -(-max(e - b for b, e in all_slices(n_rays, self.R)) // G)
        self._tile = None

    def sub_range(self):
        """Rays of the frame this rank fuses: [begin, end)."""
        b, e = self.ray_range
        return min(b + self.g * self.sub, e), min(b + (self.g + 1) * self.sub, e)

    def exchange(self, raws, zs, ms):
        """(1, K/G, n_r, Ns, 4), (1, K/G, n_r, Ns), (1, K/G, n_r, Ns) of MY volumes over the whole ray slice ->
        the same three for ALL K volumes (in order) over my ray sub-slice."""
        kl, n_r, Ns = raws.shape[1], raws.shape[2], raws.shape[3]
        G, sub = self.G, self.sub
        if G == 1:
            return raws, zs, ms
        # persistent buffers (torch.zeros / .contiguous() / torch.cat on every step are a visible cost at
        # 0.8 ms frames): the padded pack (its padding rows are zeroed once and never written), the peer-major send and
        # receive blocks and the three outputs live as long as the shapes do; the returned tensors are valid until the
        # next exchange() of this object (the blend consumes them inside the step)
        key = (kl, n_r, Ns, raws.device)
        if getattr(self, "_xbuf", None) is None or self._xbuf[0] != key:
            dev = raws.device
            pack = torch.zeros(kl, G * sub, Ns, 6, device=dev, dtype=torch.float32)
            send = torch.empty(G, kl, sub, Ns, 6, device=dev, dtype=torch.float32)
            b, e = self.sub_range()
            outs = (torch.empty(1, self.K, e - b, Ns, 4, device=dev), torch.empty(1, self.K, e - b, Ns, device=dev),
                    torch.empty(1, self.K, e - b, Ns, device=dev))
            self._xbuf = (key, pack, send, torch.empty_like(send), outs)
        _, pack, send, recv, outs = self._xbuf
        pack[:, :n_r, :, :4] = raws[0]
        pack[:, :n_r, :, 4] = zs[0]
        pack[:, :n_r, :, 5] = ms[0]
        send.copy_(pack.view(kl, G, sub, Ns, 6).transpose(0, 1))              # (G peers, kl, sub, Ns, 6)
        dist.all_to_all_single(recv, send, group=self.group)
        # recv[g'] = volumes {g', g' + G, ...} of peer g' for my sub-slice -> volume order k = j * G + g'
        # ((kl, G, sub, Ns, 6) view of the receive block: row (j, g') is volume k = j * G + g')
        n_sub = outs[0].shape[2]
        v = recv.permute(1, 0, 2, 3, 4)[:, :, :n_sub]
        outs[0][0].view(kl, G, n_sub, Ns, 4).copy_(v[..., :4])
        outs[1][0].view(kl, G, n_sub, Ns).copy_(v[..., 4])
        outs[2][0].view(kl, G, n_sub, Ns).copy_(v[..., 5])
        return outs

    def gather_tiles(self, rgb, depth):
        """Fused (1, n_sub, 3), (1, n_sub) of every rank's sub-slice -> the frame (n_rays, 4)."""
        n = rgb.shape[-2]
        if self._tile is None or self._tile[0].device != rgb.device:
            self._tile = (torch.zeros(self.sub_max, 4, device=rgb.device), torch.empty(self.world * self.sub_max, 4, device=rgb.device))
        send, recv = self._tile
        send[:n, :3] = rgb.reshape(n, 3)
        send[:n, 3] = depth.reshape(n)
        dist.all_gather_into_tensor(recv, send)
        parts = recv.view(self.world, self.sub_max, 4)
        # the frame buffer is persistent too: valid until the next gather_tiles() of this object
        if getattr(self, "_frame", None) is None or self._frame.device != rgb.device:
            self._frame = torch.empty(self.n_rays, 4, device=rgb.device)
            self._spans = []
            at = 0
            for r in range(self.R):
                b, e = ray_slice(self.n_rays, self.R, r)
                sub = -(-(e - b) // self.G)
                for g in range(self.G):
                    lo, hi = min(b + g * sub, e), min(b + (g + 1) * sub, e)
                    self._spans.append((r * self.G + g, at, hi - lo))
                    at += hi - lo
            assert at == self.n_rays
            self._even = all(n_ == self.sub_max for _, _, n_ in self._spans)
        if self._even:                       # every sub-slice is full: the gathered block IS the frame
            return parts.reshape(self.n_rays, 4)
        for rank_, at, n_ in self._spans:
            if n_:
                self._frame[at:at + n_].copy_(parts[rank_, :n_])
        return self._frame
