"""Optional per-kernel timing with HIP events on the stream the kernels run on
(torch's current stream).  Off by default; bench.py switches it on so the
roofline numbers come from the timed region itself."""
import contextlib

import torch

enabled = False
only = None          # optional tuple of name prefixes: time just these kernels (two event records cost ~10 us of host time)
_records = {}


def reset():
    _records.clear()


@contextlib.contextmanager
def region(name):
    if not enabled or (only is not None and not name.startswith(only)):
        yield
        return
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    try:
        yield
    finally:
        e.record()
        _records.setdefault(name, []).append((s, e))


def summary():
    """name -> (launches, mean ms, min ms); call after a device synchronize."""
    out = {}
    for name, evs in _records.items():
        ms = [s.elapsed_time(e) for s, e in evs]
        out[name] = (len(ms), sum(ms) / len(ms), min(ms))
    return out
