"""Optional per-kernel timing with HIP events on the stream the kernels run on
(torch's current stream).  Off by default; bench.py switches it on so the
roofline numbers come from the timed region itself.

Two kinds of bracket:
* eager launches: a pair of events per launch, read after a device synchronize (`summary`);
* launches captured into a HIP graph (`framegraph.FrameGraph(events=True)`): ONE pair per captured launch, recorded
  as event-record nodes of the graph (`bmv_event_record`, csrc/timing.hip: torch refuses event records during
  capture on ROCm).  Every replay re-stamps the pair; `collect()` -- after a synchronize, before the next replay --
  appends the durations of the replay that just finished.
"""
import contextlib
import ctypes

import torch

enabled = False
only = None          # optional tuple of name prefixes: time just these kernels (two event records cost ~10 us of host time)
_records = {}        # name -> [(start, end)] torch events of eager launches
_bound = {}          # name -> [(start, end)] events bound to a launch's own dispatch
_no_bind = set()     # names whose launch path does not take bound events
_graph_pairs = {}    # name -> [(start, end)] events living in captured graphs
_graph_us = {}       # name -> [us] collected from the graph pairs


class _Event:
    """hipEvent_t through the C-ABI (works under stream capture, see the module docstring)."""

    def __init__(self):
        from . import _lib
        self._lib = _lib
        self.h = ctypes.c_void_p()
        _lib.check(_lib.load().bmv_event_create(ctypes.byref(self.h)), "bmv_event_create")

    def record(self):
        self._lib.check(self._lib.load().bmv_event_record(self.h, self._lib.stream()), "bmv_event_record")

    def elapsed_us(self, end):
        us = ctypes.c_float()
        self._lib.check(self._lib.load().bmv_event_elapsed_us(self.h, end.h, ctypes.byref(us)), "bmv_event_elapsed_us")
        return us.value

    def __del__(self):
        try:
            self._lib.load().bmv_event_destroy(self.h)
        except Exception:
            pass


def reset():
    """Forget the collected durations (the event pairs of live graphs stay: they belong to the graphs)."""
    _records.clear()
    _bound.clear()
    _graph_us.clear()


def forget_graph_events():
    """Stop COLLECTING from the event pairs of earlier captures.  The pairs themselves stay alive as long as the
    FrameGraph that recorded them does (`graph_pairs_snapshot` -> FrameGraph.event_pairs): a replay of such a graph
    still stamps valid events."""
    _graph_pairs.clear()
    _graph_us.clear()


def graph_pairs_snapshot():
    """Every (start, end) pair recorded into a capture so far (the caller -- FrameGraph -- keeps the ones it added)."""
    return [p for pairs in _graph_pairs.values() for p in pairs]


@contextlib.contextmanager
def region(name, bind=False):
    """`bind=True` (plane sweeps): outside a capture the pair is BOUND to the kernel's own dispatch
    (bmv_bind_next_launch -> hipExtLaunchKernelGGL: the events read the kernel's begin and end) instead of recorded
    around the launch; a launch path that does not take them falls back to the recorded pair from then on."""
    if not enabled or (only is not None and not name.startswith(only)):
        yield
        return
    if bind and name not in _no_bind and not torch.cuda.is_current_stream_capturing():
        from . import _lib
        lib = _lib.load()
        s, e = _Event(), _Event()
        _lib.check(lib.bmv_bind_next_launch(s.h, e.h), "bmv_bind_next_launch")
        try:
            yield
        finally:
            if lib.bmv_launch_events_pending():
                _no_bind.add(name)                       # this shape runs another kernel: recorded pairs next time
            else:
                _bound.setdefault(name, []).append((s, e))
        return
    if torch.cuda.is_current_stream_capturing():
        s, e = _Event(), _Event()
        s.record()
        try:
            yield
        finally:
            e.record()
            _graph_pairs.setdefault(name, []).append((s, e))
        return
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    try:
        yield
    finally:
        e.record()
        _records.setdefault(name, []).append((s, e))


def collect():
    """Durations of the in-graph brackets for the replay that just completed (call after a synchronize)."""
    if not enabled:
        return
    for name, pairs in _graph_pairs.items():
        if only is not None and not name.startswith(only):
            continue
        _graph_us.setdefault(name, []).extend(s.elapsed_us(e) for s, e in pairs)


def summary():
    """name -> (launches, mean ms, min ms); call after a device synchronize."""
    out = {}
    names = set(_records) | set(_graph_us) | set(_bound)
    for name in names:
        ms = ([s.elapsed_time(e) for s, e in _records.get(name, [])] + [u * 1e-3 for u in _graph_us.get(name, [])]
              + [s.elapsed_us(e) * 1e-3 for s, e in _bound.get(name, [])])
        if ms:
            out[name] = (len(ms), sum(ms) / len(ms), min(ms))
    return out
