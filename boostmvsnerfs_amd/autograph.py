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
  bracket: the cost is charged to the iteration that triggers it), later ones replay;
* **forward never writes its inputs and never hands out memory it will overwrite** (the reference's forward does
  neither): the graph is captured on PRIVATE copies of the batch's tensors, every call copies the caller's tensors into
  them with one multi-tensor launch (23 MB at 512x640: what an unchanged run.py loop, which hands over new device
  tensors every frame, needs anyway), and the outputs are copies of the graph's static outputs (8 MB).  Two opt-ins
  trade that contract for the copies, attribute by attribute on the network:
    - `net.resident_inputs = True`: the caller DECLARES its batch resident -- the graph is captured on the caller's own
      tensors and a call with those very tensors copies nothing (in-place edits of them are picked up); a call with
      other tensors of the same shapes is copied INTO THE DECLARED BATCH'S tensors, which is what the declaration
      permits;
    - `net.alias_outputs = True`: the returned tensors are the graph's static outputs, valid until the next forward of
      this network (the reference's evaluators consume a frame before the next one is rendered);
* training, CPU tensors, a batch tensor that requires grad (pose refinement: the outputs must carry an autograd graph),
  a forward that is itself being captured, and `BMV_AUTOGRAPH=0` take the eager path.

Not noticed: a parameter whose storage is swapped through `p.data = ...` (no version bump, no hook); call
`net._autograph.invalidate()` after such an edit.  `load_state_dict` (also `assign=True`) and `.to()` invalidate.
"""
from __future__ import annotations

import os
from operator import attrgetter

import torch

_version_of = attrgetter("_version")

ENABLED = os.environ.get("BMV_AUTOGRAPH", "1") != "0"
MAX_GRAPHS = int(os.environ.get("BMV_AUTOGRAPH_MAX", "4"))


def _built(v):
    return getattr(v, "_bmv_built_rays", False)


def _copy_many(dsts, srcs):
    """dst[i] <- src[i] on the current stream: one multi-tensor launch per dtype group.  (Measured against it in round 4
    and dropped: an own single-launch copy kernel through ctypes -- +34 us per frame where this costs +22, the host
    side of the call is what counts -- and the two copies as nodes of the frame's graph re-targeted per replay with
    hipGraphExecKernelNodeSetParams -- +66 us: an updated executable graph is re-prepared at its next launch.)"""
    if not dsts:
        return
    groups = {}
    for d, s in zip(dsts, srcs):
        groups.setdefault((d.dtype, s.dtype), ([], []))
        g = groups[(d.dtype, s.dtype)]
        g[0].append(d)
        g[1].append(s)
    for d, s in groups.values():
        torch._foreach_copy_(d, s)


class AutoGraph:
    def __init__(self, net):
        """`net` provides `_forward_checked(batch) -> dict` (the eager forward) and `_autograph_key(batch)` (what else a
        captured frame is specialised to, e.g. the selected triplets; or None)."""
        self.net = net
        self.entries = {}          # key -> {"fg", "static", "hits", ...}
        self.seen = {}             # key -> number of eager calls so far
        self._tensors = None
        self._last = None          # resident inputs: ((ids of the batch's values, parameter version, shard, extra), entry)
        self._hot = None           # private copies: the entry of the last replay (fast path of _hot_call)
        self.epoch = 0             # bumped by Module._apply (.to() / .cuda() replace storage) and load_state_dict
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
        self._hot = None
        self.entries.clear()
        self.seen.clear()

    def _param_version(self):
        # in-place updates (optimiser steps, manual edits) bump `_version`; storage moves and replaced Parameter objects
        # go through Module._apply / load_state_dict -> invalidate(), which drops the cached list.  ~5 us for the 184
        # tensors of an ENeRF network (sum / map run in C)
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
        probe = None
        grad = torch.is_grad_enabled()
        for v in batch.values():
            if torch.is_tensor(v):
                if probe is None:
                    probe = v
                if grad and v.requires_grad:
                    return False               # the outputs must carry an autograd graph back to this tensor
        if grad and any(p.requires_grad for p in self.net.parameters()):
            return False
        if probe is None or not probe.is_cuda:
            return False
        return not torch.cuda.is_current_stream_capturing()

    # ------------------------------------------------------------------ call
    def _hot_call(self, batch):
        """The steady state of a loop that hands over other tensors every frame (run.py): the batch has the structure of
        the last replayed private-copy entry -> the input copy goes to the GPU FIRST (it is harmless whatever the key
        checks say: the static inputs are private), the host-side checks run under it."""
        e = self._hot
        st = e["static"]
        srcs = []
        n = 0
        for v in batch.values():
            if torch.is_tensor(v) and not _built(v):
                n += 1
        if n != e["n_tensors"]:
            return None
        for k in e["names"]:
            v = batch.get(k)
            s = st[k]
            if v is None or v.shape != s.shape or v.dtype != s.dtype or v.device != s.device:
                return None
            srcs.append(v)
        _copy_many(e["dsts"], srcs)
        if (e["version"] != self._param_version() or e["shard"] != self._shard()
                or e["extra"] != self.net._autograph_key(batch) or self.entries.get(e["key"]) is not e):
            return None
        self.stats["copies"] += len(srcs)
        self.stats["replays"] += 1
        out = e["fg"].replay()
        e["hits"] += 1
        for k, v in e["added"].items():
            batch[k] = v
        if getattr(self.net, "alias_outputs", False):
            return dict(out)
        return self._fresh_outputs(e, out)

    def __call__(self, batch):
        resident = bool(getattr(self.net, "resident_inputs", False))
        if not resident and self._hot is not None:
            res = self._hot_call(batch)
            if res is not None:
                return res
        version = self._param_version()
        extra = self.net._autograph_key(batch)
        ident = None
        e = None
        if resident:
            # fast path: the very tensor objects of the last replay (the declared resident batch) -> nothing to copy
            ident = (tuple(map(id, batch.values())), version, self._shard(), extra)
            last = self._last
            if last is not None and last[0] == ident:
                e = last[1]
        if e is None:
            key = (self._shapes(batch), self._shard(), extra, resident)
            e = self.entries.get(key)
            if e is not None and e["version"] != version:
                del self.entries[key]
                e = None
            if e is None:
                n = self.seen.get(key, 0)
                self.seen[key] = n + 1
                if n == 0:                              # a shape seen once is not worth a capture
                    if len(self.seen) > 64:
                        self.seen.clear()
                    self.stats["eager"] += 1
                    return self.eager_forward(batch)
                e = self._capture(key, batch, version, resident)
            copied = self._refresh(e, batch)
            if not resident:
                e["key"], e["shard"], e["extra"] = key, key[1], extra
                self._hot = e
            if resident:
                # these objects ARE the static inputs if nothing had to be copied: remember them for the fast path
                self._last = (ident, e) if not copied else None
        self.stats["replays"] += 1
        out = e["fg"].replay()
        e["hits"] += 1
        for k, v in e["added"].items():             # keys the eager forward adds to the batch (rays built on the device)
            batch[k] = v
        if resident and self._last is not None and e["added"] and len(ident[0]) != len(batch):
            self._last = ((tuple(map(id, batch.values())),) + ident[1:], e)       # (the batch just gained those keys)
        if getattr(self.net, "alias_outputs", False):
            return dict(out)
        return self._fresh_outputs(e, out)

    @staticmethod
    def _fresh_outputs(e, out):
        names = e.get("out_names")
        if names is None:
            names = e["out_names"] = [k for k, v in out.items() if torch.is_tensor(v)]
        srcs = [out[k] for k in names]
        dsts = [torch.empty_like(s) for s in srcs]
        _copy_many(dsts, srcs)
        res = dict(out)
        res.update(zip(names, dsts))
        return res

    def _capture(self, key, batch, version, resident):
        from .framegraph import FrameGraph
        if len(self.entries) >= MAX_GRAPHS:          # every graph owns a private memory pool: keep a few
            victim = min(self.entries, key=lambda k: self.entries[k]["hits"])
            del self.entries[victim]
        # static inputs of the graph: private copies (default), or -- declared resident -- the caller's own tensors, kept
        # alive by this reference.  Python entries ('meta') are read at capture time only: anything of theirs that
        # changes the frame is part of the key
        reads = getattr(self.net, "_autograph_inputs", lambda b: None)(batch)     # None: every tensor of the batch
        static = {}
        for k, v in batch.items():
            if _built(v):
                continue
            if torch.is_tensor(v) and not resident:
                # tensors the frame does not read stay the caller's (never copied, never written)
                static[k] = v.clone() if (reads is None or k in reads) else v
            else:
                static[k] = v
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
        names = [k for k, v in static.items() if torch.is_tensor(v) and (resident or reads is None or k in reads)]
        e = {"fg": fg, "static": static, "hits": 0, "version": version, "added": added, "resident": resident,
             "names": names, "dsts": [static[k] for k in names], "one_dtype": len({static[k].dtype for k in names}) == 1,
             "n_tensors": sum(1 for v in static.values() if torch.is_tensor(v))}
        self.entries[key] = e
        return e

    def _refresh(self, e, batch):
        """The caller's tensors into the graph's static inputs; returns the number of tensors copied."""
        static = e["static"]
        dsts, srcs = [], []
        for k in e["names"]:
            v = batch.get(k)
            if v is None or not torch.is_tensor(v):
                continue
            s = static[k]
            if s is v or (s.data_ptr() == v.data_ptr() and s.stride() == v.stride()):
                continue                               # (only possible when the capture ran on the caller's tensors)
            dsts.append(s)
            srcs.append(v)
        _copy_many(dsts, srcs)
        self.stats["copies"] += len(dsts)
        return len(dsts)
