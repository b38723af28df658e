"""Do external HIP events recorded inside a captured graph (hipEventRecordWithFlags(.., hipEventRecordExternal): torch
refuses them on ROCm, so straight through the runtime) time a kernel correctly?  Compares the in-graph bracket of a
kernel of known length with ordinary events around the same kernel launched eagerly."""
import ctypes
import time

import torch

hip = ctypes.CDLL("libamdhip64.so.7")       # the runtime torch already loaded (same SONAME)
dev = "cuda"
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)
a = torch.randn(4096, 4096, device=dev)


class Ev:
    def __init__(self):
        self.h = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(self.h)) == 0

    def record(self):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if not torch.cuda.is_current_stream_capturing():
            assert hip.hipEventRecord(self.h, s) == 0
            return
        # capturing: add an event-record node behind the capture's current leaves and make later work depend on it
        status, cid, graph = ctypes.c_int(), ctypes.c_ulonglong(), ctypes.c_void_p()
        deps, nd = ctypes.POINTER(ctypes.c_void_p)(), ctypes.c_size_t()
        r = hip.hipStreamGetCaptureInfo_v2(s, ctypes.byref(status), ctypes.byref(cid), ctypes.byref(graph),
                                           ctypes.byref(deps), ctypes.byref(nd))
        assert r == 0 and status.value == 1, (r, status.value)
        node = ctypes.c_void_p()
        r = hip.hipGraphAddEventRecordNode(ctypes.byref(node), graph, deps, nd, self.h)
        assert r == 0, r
        r = hip.hipStreamUpdateCaptureDependencies(s, ctypes.byref(node), ctypes.c_size_t(1), 1)   # 1 = set
        assert r == 0, r

    def elapsed_us(self, other):
        ms = ctypes.c_float()
        r = hip.hipEventElapsedTime(ctypes.byref(ms), self.h, other.h)
        return ms.value * 1e3 if r == 0 else -r


def body(ev=None):
    b = a @ a
    if ev:
        ev[0].record()
    torch.mul(x, 2.0, out=y)            # 256 MB read + 256 MB write
    if ev:
        ev[1].record()
    return b @ a


for _ in range(3):
    body()
torch.cuda.synchronize()
s, e = Ev(), Ev()
ms = []
for _ in range(8):
    body((s, e))
    torch.cuda.synchronize()
    ms.append(round(s.elapsed_us(e), 1))
print("eager bracket us:", ms)

xs, xe = Ev(), Ev()
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    body()
    torch.cuda.synchronize()
    g.capture_begin()
    out = body((xs, xe))
    g.capture_end()
torch.cuda.synchronize()
ms = []
for _ in range(8):
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms.append((round(xs.elapsed_us(xe), 1), round(dt * 1e6)))
print("graph bracket us (event, wall of whole replay):", ms)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    g2.capture_begin()
    out = body()
    g2.capture_end()
torch.cuda.synchronize()
w = []
for _ in range(8):
    t0 = time.perf_counter()
    g2.replay()
    torch.cuda.synchronize()
    w.append(round((time.perf_counter() - t0) * 1e6))
print("graph without events, wall us:", w)
