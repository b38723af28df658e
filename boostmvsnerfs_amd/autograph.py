"""Self-capturing inference forward: `network(batch)` replays a HIP graph of the frame.

The drop-in boundary is `Network.forward(batch)` (run.py:113-123 calls nothing else).  Issued eagerly, the ~42 launches
of a 512x640 frame cost the host as long as the GPU needs to run them, so the reference's own timing bracket around an
unchanged `network(batch)` was host-bound (round 2: 323 Mray/s eager against 353 replayed through a separate harness).
`AutoGraph` moves the replay behind `forward` itself:

* a call is keyed by what a captured frame is specialised to beyond tensor VALUES: the shapes / dtypes / devices of the
  batch's tensors, the parameters' versions (an optimiser step, `load_state_dict`, `.to()` all invalidate), the ray /
  volume shard of the network and, for the K-volume networks, the cost-volume triplets `view_selection.json` selects
  for the batch's targets;
* the FIRST call with a key runs eagerly; the second captures (`framegraph.FrameGraph`, inside the caller's timing
  bracket: the cost is charged to the iteration that triggers it) ON THE CALLER'S OWN TENSORS, later ones replay.  A
  caller that keeps its batch resident pays nothing per frame; one that hands over new tensors every frame pays one
  in-place device copy per tensor into the captured buffers (what evaluate.py's refresh did from outside);
* outputs are the graph's static tensors: valid until the next forward of this network (the reference's evaluators
  consume a frame before the next one is rendered); `BMV_AUTOGRAPH_CLONE=1` hands out copies instead;
* training, CPU tensors, a forward that is itself being captured, and `BMV_AUTOGRAPH=0` take the eager path.
"""
from __future__ import annotations

import os
from operator import attrgetter

import torch

_version_of = attrgetter("_version")

ENABLED = os.environ.get("BMV_AUTOGRAPH", "1") != "0"
CLONE_OUTPUTS = os.environ.get("BMV_AUTOGRAPH_CLONE", "0") == "1"
MAX_GRAPHS = int(os.environ.get("BMV_AUTOGRAPH_MAX", "4"))


def _built(v):
    return getattr(v, "_bmv_built_rays", False)


class AutoGraph:
    def __init__(self, net):
        """`net` provides `_forward_checked(batch) -> dict` (the eager forward) and `_autograph_key(batch)` (what else a
        captured frame is specialised to, e.g. the selected triplets; or None)."""
        self.net = net
        self.entries = {}          # key -> {"fg", "static", "hits"}
        self.seen = {}             # key -> number of eager calls so far
        self._tensors = None
        self._last = None          # ((ids of the batch's values, parameter version, shard, extra key), entry) of the last replay
        self.epoch = 0             # bumped by Module._apply (.to() / .cuda() replace storage)
        self.stats = {"eager": 0, "captures": 0, "replays": 0, "copies": 0}

    def eager_forward(self, batch):
        return self.net._forward_checked(batch)

    def __deepcopy__(self, memo):        # captured graphs belong to one module instance (copy.deepcopy, DDP wrappers)
        import copy
        return AutoGraph(copy.deepcopy(self.net, memo))

    def __getstate__(self):
        return {"net": self.net}

    def __setstate__(self, state):
        self.__init__(state["net"])

    # ------------------------------------------------------------------ keys
    def invalidate(self):
        self.epoch += 1
        self._tensors = None
        self._last = None
        self.entries.clear()
        self.seen.clear()

    def _param_version(self):
        # in-place updates (optimiser steps, load_state_dict, manual edits) bump `_version`; storage moves go through
        # Module._apply -> invalidate().  ~5 us for the 184 tensors of an ENeRF network (sum / map run in C)
        if self._tensors is None:
            self._tensors = list(self.net.parameters()) + list(self.net.buffers())
        return self.epoch + sum(map(_version_of, self._tensors))

    def _shard(self):
        net = self.net
        return getattr(net, "ray_range", None), tuple(getattr(net, "volume_ids", None) or ()) or None

    @staticmethod
    def _shapes(batch):
        # (rays an earlier forward built on the device are not inputs: they are rebuilt from the camera inside the frame)
        return tuple((k, tuple(v.shape), v.dtype, v.device.index) for k, v in sorted(batch.items())
                     if torch.is_tensor(v) and not _built(v))

    def usable(self, batch):
        if not ENABLED or self.net.training:
            return False
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            return False
        probe = next((v for v in batch.values() if torch.is_tensor(v)), None)
        if probe is None or not probe.is_cuda:
            return False
        return not torch.cuda.is_current_stream_capturing()

    # ------------------------------------------------------------------ call
    def __call__(self, batch):
        version = self._param_version()
        extra = self.net._autograph_key(batch)
        # fast path: the very tensor objects of the last replay (a resident batch) -> same key, nothing to copy
        ident = (tuple(map(id, batch.values())), version, self._shard(), extra)
        last = self._last
        if last is not None and last[0] == ident:
            e = last[1]
        else:
            key = (self._shapes(batch), ident[2], extra)
            e = self.entries.get(key)
            if e is not None and e["version"] != version:
                del self.entries[key]
                e = None
            copies = self.stats["copies"]
            if e is None:
                n = self.seen.get(key, 0)
                self.seen[key] = n + 1
                if n == 0:                              # a shape seen once is not worth a capture
                    if len(self.seen) > 64:
                        self.seen.clear()
                    self.stats["eager"] += 1
                    return self.eager_forward(batch)
                e = self._capture(key, batch, version)
            else:
                self._refresh(e["static"], batch)
            # these objects ARE the static inputs if nothing had to be copied: remember them for the fast path
            self._last = (ident, e) if copies == self.stats["copies"] else None
        self.stats["replays"] += 1
        out = e["fg"].replay()
        e["hits"] += 1
        for k, v in e["added"].items():             # keys the eager forward adds to the batch (rays built on the device)
            batch[k] = v
        if self._last is not None and e["added"] and len(ident[0]) != len(batch):
            self._last = ((tuple(map(id, batch.values())),) + ident[1:], e)       # (the batch just gained those keys)
        if CLONE_OUTPUTS:
            return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
        return dict(out)

    def _capture(self, key, batch, version):
        from .framegraph import FrameGraph
        if len(self.entries) >= MAX_GRAPHS:          # every graph owns a private memory pool: keep a few
            victim = min(self.entries, key=lambda k: self.entries[k]["hits"])
            del self.entries[victim]
        # the caller's tensors ARE the static inputs (kept alive by this reference); python entries ('meta') are read
        # at capture time only -- anything of theirs that changes the frame is part of the key
        static = {k: v for k, v in batch.items() if not _built(v)}
        added = {}

        def run(b):
            # forward may ADD keys to the batch (rays built on the device from the target camera): every call gets a
            # fresh shallow copy, so that such tensors are rebuilt inside the captured frame instead of being baked in
            bb = dict(b)
            out = self.eager_forward(bb)
            for k, v in bb.items():
                if k not in b:
                    added[k] = v
            return out
        with torch.no_grad():
            fg = FrameGraph(run, static, cut=None)
        self.stats["captures"] += 1
        e = {"fg": fg, "static": static, "hits": 0, "version": version, "added": added}
        self.entries[key] = e
        return e

    def _refresh(self, static, batch):
        for k, v in batch.items():
            if not torch.is_tensor(v) or _built(v):
                continue
            s = static.get(k)
            if s is None or s is v or (s.data_ptr() == v.data_ptr() and s.stride() == v.stride()):
                continue
            s.copy_(v)
            self.stats["copies"] += 1
