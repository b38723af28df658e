"""HIP-graph replay of `Network.forward` for a resident batch.

A frame is ~45 short launches; issued one by one from Python the host needs about as long as the GPU
(and cannot run far enough ahead for the two-stream overlap of networks/enerf/network.py to happen).
`FrameGraph` captures one forward pass -- including its second-stream fork/join -- into HIP graphs and
replays them: the host cost of a frame drops to three calls.

The capture is CUT at plane-sweep launches (`cut=-1`: the last one, the level-1 sweep; `cut="all"`: every
sweep issued on the capture stream): the graphs hold everything between them and the sweeps themselves
stay ordinary launches, so HIP events on the stream can time exactly those kernels inside a timed region
(bench.py's `roofline`).  `cut=None` captures the frame as one graph.

`events=True`: the kernels `ktimer` is set to time are bracketed INSIDE the graphs by event-record nodes
(csrc/timing.hip); `ktimer.collect()` after a synchronize reads the brackets of the replay that just finished.  With
`cut="all"` as well (bench.py's bracketed frames) the sweeps are ordinary launches between the graphs whose events are
bound to their own dispatch (`ktimer.region(bind=True)`: they read the kernel's begin and end), the renderer keeps its
in-graph bracket.

The batch tensors are the graph's static inputs: refresh them in place (`copy_`) between replays.  The
returned dict holds the static output tensors of the captured pass.
"""
from __future__ import annotations

import torch

from . import ktimer, ops


class FrameGraph:
    def __init__(self, net, batch, cut=-1, warmup=2, events=False):
        # (a network with a self-capturing forward -- autograph.AutoGraph -- is captured through its eager entry point)
        self.net, self.batch = getattr(net, "_forward_checked", net), batch
        self.cut = cut
        self.events = events
        self.graphs = []
        self.sweeps = []            # (impl, args, kwargs) of the eager sweep launched after graph i
        self.out = None
        self.event_pairs = []       # the event-record nodes' events: owned by this graph (ktimer may forget them)
        before = set(map(id, ktimer.graph_pairs_snapshot()))
        self._capture(warmup)
        self.event_pairs = [p for p in ktimer.graph_pairs_snapshot() if id(p) not in before]

    # ------------------------------------------------------------------ capture
    def _count_sweeps(self):
        calls = []

        def hook(impl, args, kwargs):
            calls.append(torch.cuda.current_stream())
            return None
        ops.sweep_hook = hook
        try:
            with torch.no_grad():
                self.net(self.batch)
        finally:
            ops.sweep_hook = None
        return calls

    def _capture(self, warmup):
        # graphs of earlier frames (this module's, or the ones Network.forward captures of itself) may be garbage by
        # now: destroy them BEFORE the capture starts and keep the collector out of it (a hipGraphExec destroyed by a
        # collection in the middle of a multi-stream capture crashed the process; torch.cuda.graph() collects first too)
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            self._capture_inner(warmup)
        finally:
            if gc_was:
                gc.enable()

    def _capture_inner(self, warmup):
        was_enabled, ktimer.enabled = ktimer.enabled, False
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(stream), torch.no_grad():
                for _ in range(warmup):              # allocator pools, packed weights, side stream: all warm
                    self.net(self.batch)
                cut_index = set()
                if self.cut is not None:
                    calls = self._count_sweeps()
                    if self.cut == "all":
                        wanted = range(len(calls))
                    else:
                        wanted = [self.cut if self.cut >= 0 else len(calls) + self.cut]
                    # only a launch on the capture stream itself can split the capture
                    cut_index = {i for i in wanted if 0 <= i < len(calls) and calls[i] == stream}
                torch.cuda.synchronize()
                seen = [0]
                g_a = torch.cuda.CUDAGraph()
                self.graphs = [g_a]

                def hook(impl, args, kwargs):
                    i = seen[0]
                    seen[0] += 1
                    if i not in cut_index:
                        return None
                    self.graphs[-1].capture_end()
                    res = impl(*args, **kwargs)                      # eager, on static buffers
                    self.sweeps.append((impl, args, {**kwargs, "out": res}))
                    g_b = torch.cuda.CUDAGraph()
                    self.graphs.append(g_b)
                    g_b.capture_begin(pool=g_a.pool(), capture_error_mode="thread_local")
                    return res
                ops.sweep_hook = hook
                try:
                    g_a.capture_begin(capture_error_mode="thread_local")
                    ktimer.enabled = self.events          # brackets become event-record nodes of the graph
                    try:
                        if self.events:
                            # what a bracket reads with nothing inside: the cost of the two records themselves, which
                            # every bracket of this graph includes (bench.py reports it next to the durations)
                            with ktimer.region("empty_bracket"):
                                pass
                        self.out = self.net(self.batch)
                    finally:
                        ktimer.enabled = False
                        self.graphs[-1].capture_end()
                finally:
                    ops.sweep_hook = None
        finally:
            ktimer.enabled = was_enabled
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ replay
    @property
    def sweep_args(self):
        """The last eager sweep (None if the frame is one graph)."""
        return self.sweeps[-1] if self.sweeps else None

    def replay(self):
        self.graphs[0].replay()
        for (impl, args, kwargs), g in zip(self.sweeps, self.graphs[1:]):
            impl(*args, **kwargs)                             # times itself through ktimer when enabled
            g.replay()
        return self.out
