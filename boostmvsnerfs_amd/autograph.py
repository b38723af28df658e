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
  them with one multi-tensor launch, and the outputs are copies of the graph's static outputs.  The LARGE tensors do
  not move at all: the kernels that read the images and the rays, and the renderer that writes rgb / depth / weights,
  take those pointers from a device table when they RUN (`ops.PtrTable`, include/bmv.h `bmv_defer_pointer`); a call
  points the table at the caller's tensors and at freshly allocated outputs (one 1-workgroup launch) instead of
  copying 22 MB in and 8 MB out.  Which inputs may be deferred is the network's statement
  (`_autograph_deferrable`); whether the captured frame really reads them ONLY through the table is CHECKED at capture
  time: a second replay with the private copies and the static outputs poisoned and the table pointing elsewhere must
  reproduce the first one bit for bit, else the entry falls back to copies.  Two opt-ins
  trade the contract for the remaining copies, attribute by attribute on the network:
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

from operator import attrgetter

import ctypes as C

import torch

from . import _lib, ops, switches

_version_of = attrgetter("_version")

# BMV_AUTOGRAPH (self-capturing forward), BMV_AUTOGRAPH_DEFER (large inputs / outputs through a pointer table, no
# copies), BMV_AUTOGRAPH_RING (the table fed by the frame's own first node from a host ring, no launch) and
# BMV_AUTOGRAPH_MAX (graphs kept per network) are read when a call / a capture needs them: `switches.set()` after import
# takes effect on the next call (entries captured before keep the form they were captured in)


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
    def check_faults(self):
        """Sequence faults of the captured frames' feed rings (ops.FeedRing.state[1]: a frame that ran on a message that
        was not its own): raises.  Reads the device -- call where the caller is synchronised anyway (evaluate(), bench's
        timed loop, invalidate())."""
        for e in self.entries.values():
            ring = e.get("ring")
            if ring is not None:
                ring.raise_on_faults()

    def invalidate(self):
        if self.entries and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            self.check_faults()           # the entries are about to go: what they counted must not go unseen
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
        if switches.get("BMV_AUTOGRAPH") == 0 or self.net.training:
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
        the last replayed private-copy entry and passes the key checks -> its tensors are fed to the captured frame and
        the graph is replayed (nothing else is launched)."""
        e = self._hot
        # the batch has the keys of the captured one (plus, possibly, the keys an earlier forward added: built rays) ...
        if len(batch) not in e["n_items"] or not e["keys"].issuperset(batch):
            return None
        # ... and the tensors the frame reads have the captured shapes / types
        for k, shape, dtype, device in e["sig"]:
            v = batch.get(k)
            if v is None or v.shape != shape or v.dtype is not dtype or v.device != device:
                return None
        if (e["version"] != self._param_version() or e["shard"] != self._shard()
                or e["extra"] != self.net._autograph_key(batch) or self.entries.get(e["key"]) is not e):
            return None
        fresh = self._feed(e, batch)        # (after the checks: a ring message posted here is read by THIS replay)
        self.stats["replays"] += 1
        out = e["fg"].replay()
        if e.get("ring") is not None:
            e["ring"].replayed()
        e["hits"] += 1
        for k, v in e["added"].items():
            batch[k] = v
        return self._results(e, out, fresh)

    @staticmethod
    def _next_outputs(e):
        """The NEXT call's output tensors, allocated while the GPU runs this frame: they have to exist before the next
        replay is launched (their addresses go into the table), and five allocations in front of the launch are ~20 us
        of host time on the critical path of a synchronized loop."""
        d = e.get("defer")
        if d is not None and d["feed"] is not None:
            # (allocated on the stream this frame runs on: the next call uses them only if it runs there too)
            e["next_out"] = (_lib.stream().value, [torch.empty_like(sbuf) for _, _, sbuf in d["out"]])

    # ------------------------------------------------------------------ inputs in, outputs out
    def _feed(self, e, batch):
        """This call's tensors into the captured frame: the deferred inputs (and the outputs' destinations) as pointers
        into the table, the small ones copied into the static inputs -- one launch (bmv_frame_feed) when the entry has
        a table, else one multi-tensor copy.  Returns the fresh output tensors by output name."""
        static = e["static"]
        d = e.get("defer")
        f = d["feed"] if d is not None else None
        if f is not None and f["ring"] is not None:
            # no launch at all: this replay's message into the host ring the frame's first node reads
            ring = f["ring"]
            values, srcs = [], []
            for k, _ in d["in"]:
                v = batch[k]
                if not v.is_contiguous():                   # rare: through the private copy after all
                    static[k].copy_(v)
                    v = static[k]
                    self.stats["copies"] += 1
                values.append(v.data_ptr())
            fresh = {}
            alias = getattr(self.net, "alias_outputs", False)
            ready = e.pop("next_out", None)
            if ready is not None:
                ready = ready[1] if ready[0] == _lib.stream().value else None
            for j, (k, _, sbuf) in enumerate(d["out"]):
                t = sbuf if alias else (ready[j] if ready is not None else torch.empty_like(sbuf))
                fresh[k] = t
                values.append(t.data_ptr())
            ok = True
            for k in e["names"]:
                v = batch[k]
                if not v.is_contiguous():
                    ok = False
                    break
                srcs.append(v.data_ptr())
            if ok:
                if ring.fast is None:
                    ring.prepare_fast(*f["ring_args"])
                ring.post_fast(values, srcs)
                self.stats["copies"] += f["m"]
            else:
                _copy_many(e["dsts"], [batch[k] for k in e["names"]])
                ring.post(f["ring_args"][0], values)
                self.stats["copies"] += len(e["names"])
            self.stats["deferred"] = self.stats.get("deferred", 0) + f["n"]
            return fresh
        if e.get("ring") is not None:
            e["ring"].post()            # (the frame's first node reads one message per replay, deferral adopted or not)
        if f is not None:
            values, src = f["values"], f["src"]
            i = 0
            for k, _ in d["in"]:
                v = batch[k]
                if not v.is_contiguous():                   # rare: through the private copy after all
                    static[k].copy_(v)
                    v = static[k]
                    self.stats["copies"] += 1
                values[i] = v.data_ptr()
                i += 1
            fresh = {}
            alias = getattr(self.net, "alias_outputs", False)
            ready = e.pop("next_out", None)
            if ready is not None:
                ready = ready[1] if ready[0] == _lib.stream().value else None
            for j, (k, _, sbuf) in enumerate(d["out"]):
                t = sbuf if alias else (ready[j] if ready is not None else torch.empty_like(sbuf))
                fresh[k] = t
                values[i] = t.data_ptr()
                i += 1
            ok = True
            for j, k in enumerate(e["names"]):
                v = batch[k]
                if not v.is_contiguous():
                    ok = False
                    break
                src[j] = v.data_ptr()
            if ok:
                rc = f["fn"](f["table"], f["n"], f["slots"], values, f["m"], src, f["dst"], f["cnt"], _lib.stream())
                if rc:
                    _lib.check(rc, "frame_feed")
                self.stats["copies"] += f["m"]
                self.stats["deferred"] = self.stats.get("deferred", 0) + f["n"]
                return fresh
            _copy_many(e["dsts"], [batch[k] for k in e["names"]])
            d["tb"].set(list(f["slots"])[:f["n"]], [values[q] for q in range(f["n"])])
            self.stats["copies"] += len(e["names"])
            return fresh
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
        if d is None:
            return None
        slots, ptrs = [], []
        for k, slot in d["in"]:
            v = batch[k]
            if not v.is_contiguous():
                static[k].copy_(v)
                v = static[k]
                self.stats["copies"] += 1
            slots.append(slot)
            ptrs.append(v.data_ptr())
        fresh = {}
        alias = getattr(self.net, "alias_outputs", False)
        for k, slot, sbuf in d["out"]:
            t = sbuf if alias else torch.empty_like(sbuf)
            fresh[k] = t
            slots.append(slot)
            ptrs.append(t.data_ptr())
        d["tb"].set(slots, ptrs)
        self.stats["deferred"] = self.stats.get("deferred", 0) + len(slots)
        return fresh

    def _results(self, e, out, fresh):
        if getattr(self.net, "alias_outputs", False):
            return dict(out)
        if fresh:
            self._next_outputs(e)
        names = e.get("out_names")
        if names is None:
            skip = set(fresh) if fresh else ()
            names = e["out_names"] = [k for k, v in out.items() if torch.is_tensor(v) and k not in skip]
        srcs = [out[k] for k in names]
        dsts = [torch.empty_like(s) for s in srcs]
        _copy_many(dsts, srcs)
        res = dict(out)
        res.update(zip(names, dsts))
        if fresh:
            res.update(fresh)
        return res

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
            before = self.stats["copies"]
            fresh = self._feed(e, batch)
            copied = self.stats["copies"] - before
            if not resident:
                e["key"], e["shard"], e["extra"] = key, key[1], extra
                self._hot = e
            if resident:
                # these objects ARE the static inputs if nothing had to be copied: remember them for the fast path
                self._last = (ident, e) if not copied else None
        else:
            fresh = None
        self.stats["replays"] += 1
        out = e["fg"].replay()
        if e.get("ring") is not None:
            e["ring"].replayed()
        e["hits"] += 1
        for k, v in e["added"].items():             # keys the eager forward adds to the batch (rays built on the device)
            batch[k] = v
        if resident and self._last is not None and e["added"] and len(ident[0]) != len(batch):
            self._last = ((tuple(map(id, batch.values())),) + ident[1:], e)       # (the batch just gained those keys)
        return self._results(e, out, fresh)

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
        if len(self.entries) >= max(1, int(switches.get("BMV_AUTOGRAPH_MAX"))):          # every graph owns a private memory pool: keep a few
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
            tb0 = ops.defer_table
            if tb0 is not None and tb0.ring is not None and torch.cuda.is_current_stream_capturing():
                tb0.ring.node(tb0)          # first node of the frame: this replay's table entries and small inputs
            out = self.eager_forward(bb)
            for k, v in bb.items():
                if k not in b:
                    added[k] = v
            tb_ = ops.defer_table
            if tb_ is not None and torch.cuda.is_current_stream_capturing():
                # the outputs no kernel writes through the table (small maps produced by torch ops) are copied to their
                # table entries by nodes of the frame's own graph: nothing is left to copy after a replay
                # (one launch for all of them; a network may have placed them earlier: ops.defer_small_outputs)
                ops.defer_small_outputs([v for v in out.values() if torch.is_tensor(v)])
            return out
        # the large inputs the network says its kernels can read through a pointer table (ops.PtrTable): registered
        # BEFORE the capture so that the wrappers defer them while the frame is captured
        tb = None
        want = () if (resident or switches.get("BMV_AUTOGRAPH_DEFER") == 0) else getattr(self.net, "_autograph_deferrable", lambda b: ())(batch)
        def_in = []
        if want:
            probe = next(v for v in static.values() if torch.is_tensor(v))
            tb = ops.PtrTable(probe.device)
            if switches.get("BMV_AUTOGRAPH_RING") != 0:
                tb.ring = ops.FeedRing(probe.device)
            for k in want:
                v = static.get(k)
                if torch.is_tensor(v) and (reads is None or k in reads) and v.is_contiguous():
                    def_in.append((k, tb.add_input(v)))
        prev = ops.defer_table
        ops.defer_table = tb
        try:
            with torch.no_grad():
                fg = FrameGraph(run, static, cut=None)
        finally:
            ops.defer_table = prev
        self.stats["captures"] += 1
        all_names = [k for k, v in static.items() if torch.is_tensor(v) and (resident or reads is None or k in reads)]
        e = {"fg": fg, "static": static, "hits": 0, "version": version, "added": added, "resident": resident,
             "all_names": all_names, "names": all_names, "dsts": [static[k] for k in all_names],
             "n_tensors": sum(1 for v in static.values() if torch.is_tensor(v)), "defer": None,
             # (the steady-state path's structural check: precomputed)
             "sig": [(k, static[k].shape, static[k].dtype, static[k].device) for k in all_names],
             "keys": frozenset(batch) | frozenset(added), "n_items": (len(frozenset(batch) - frozenset(added)),
                                                                       len(frozenset(batch) | frozenset(added)))}
        if tb is not None:
            e["tb"] = tb        # the graph reads the table's memory on every replay, deferral adopted or not: keep it alive
            e["ring"] = tb.ring # ... and its first node reads one ring message per replay: every replay posts one
            self._adopt_table(e, tb, def_in)
        self.entries[key] = e
        return e

    def _adopt_table(self, e, tb, def_in):
        """After the capture: which registered inputs some launch really deferred, which static outputs the renderer
        registered; the table initialised to the static tensors themselves; then the CHECK that the captured frame
        reads / writes them only through the table -- a replay with the private copies and static outputs poisoned and
        the table pointing at other memory must reproduce the first replay bit for bit.  Else: copies, as before."""
        fg, static = e["fg"], e["static"]
        def_in = [(k, slot) for k, slot in def_in if slot in tb.taken]
        out = fg.out
        def_out = []
        slots, ptrs = [], []
        for k, slot in def_in:
            slots.append(slot), ptrs.append(static[k].data_ptr())
        for ptr, (slot, t) in tb.outputs.items():
            slots.append(slot), ptrs.append(ptr)              # (every slot always points somewhere valid)
            for name, v in out.items():
                if torch.is_tensor(v) and v.data_ptr() == ptr and v.numel() == t.numel() and v.is_contiguous():
                    def_out.append((name, slot, v))
                    break
        if not slots:
            return
        tb.set(slots, ptrs)
        if not def_in and not def_out:
            return
        ring = tb.ring
        if ring is not None:
            ring.post()                       # (nothing to change: the table points at the static tensors)
        ref = {k: v.clone() for k, v in fg.replay().items() if torch.is_tensor(v)}
        if ring is not None:
            ring.replayed()
        alt_in = {k: static[k].clone() for k, _ in def_in}
        alt_out = {k: torch.empty_like(v) for k, _, v in def_out}
        for k, _ in def_in:
            static[k].fill_(float("nan")) if static[k].is_floating_point() else static[k].zero_()
        for _, _, v in def_out:
            v.fill_(float("nan")) if v.is_floating_point() else v.zero_()
        alt_slots = [s for _, s in def_in] + [s for _, s, _ in def_out]
        alt_ptrs = [alt_in[k].data_ptr() for k, _ in def_in] + [alt_out[k].data_ptr() for k, _, _ in def_out]
        if ring is not None:
            ring.post(alt_slots, alt_ptrs)    # (the check goes through the ring too)
        else:
            tb.set(alt_slots, alt_ptrs)
        got = dict(fg.replay())
        if ring is not None:
            ring.replayed()
        got.update(alt_out)
        ok = all(torch.equal(got[k], v) for k, v in ref.items())
        torch.cuda.current_stream().synchronize()
        if ring is not None and ring.faults():
            ok = False
        # (back to the static tensors either way: the table never points at memory of this function)
        for k, _ in def_in:
            static[k].copy_(alt_in[k])
        tb.set(slots, ptrs)
        self.stats["defer_checks"] = self.stats.get("defer_checks", 0) + 1
        if not ok:
            self.stats["defer_rejected"] = self.stats.get("defer_rejected", 0) + 1
            return
        gone = {k for k, _ in def_in}
        e["names"] = [k for k in e["all_names"] if k not in gone]
        e["dsts"] = [static[k] for k in e["names"]]
        d = e["defer"] = {"tb": tb, "in": def_in, "out": def_out, "feed": None}
        # table entries + the remaining (small) inputs in ONE launch (bmv_frame_feed), arguments prebuilt
        small = [static[k] for k in e["names"]]
        if len(small) <= 8 and all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= 65536 for t in small):
            n, m = len(def_in) + len(def_out), len(small)
            d["feed"] = {"n": n, "m": m, "slots": (C.c_int * max(n, 1))(*([s for _, s in def_in] + [s for _, s, _ in def_out])),
                         "values": (C.c_void_p * max(n, 1))(), "src": (C.c_void_p * max(m, 1))(),
                         "dst": (C.c_void_p * max(m, 1))(*[t.data_ptr() for t in small]),
                         "cnt": (C.c_int * max(m, 1))(*[t.numel() for t in small]), "fn": _lib.load().bmv_frame_feed,
                         "table": C.c_void_p(tb.t.data_ptr()), "ring": ring,
                         "ring_args": ([s for _, s in def_in] + [s for _, s, _ in def_out], [t.data_ptr() for t in small],
                                       [t.numel() for t in small])}
