"""Can dispatch-bound events (hipExtLaunchKernelGGL start / stop, bmv_bind_next_launch) be CAPTURED into a HIP graph?
Captures one windowed sweep with a bound pair, replays it, and reads the pair after every replay."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from boostmvsnerfs_amd import _lib, ktimer, ops

dev = "cuda"
torch.manual_seed(0)
B, S, C, Hs, Ws, D, h, w = 1, 3, 16, 128, 160, 8, 128, 160
feats = torch.randn(B, S, Hs, Ws, C, device=dev)
proj = torch.eye(3, 4, device=dev).repeat(B, S, 1, 1).contiguous()
dv = torch.rand(B, D, h, w, device=dev) + 1.0
out = ops._sweep_variance(feats, proj, dv, algo=0, channels_last=True)
torch.cuda.synchronize()
lib = _lib.load()
s, e = ktimer._Event(), ktimer._Event()
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(st):
        g.capture_begin()
        rc = lib.bmv_bind_next_launch(s.h, e.h)
        ops._sweep_variance(feats, proj, dv, algo=0, channels_last=True, out=out)
        pending = lib.bmv_launch_events_pending()
        g.capture_end()
    print("captured: bind rc", rc, "pending after launch", pending)
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        us = ctypes.c_float()
        rc = lib.bmv_event_elapsed_us(s.h, e.h, ctypes.byref(us))
        print("replay", i, "elapsed rc", rc, "us", us.value, lib.bmv_last_error().decode()[:80] if rc else "")
except Exception as ex:       # noqa: BLE001
    print("capture failed:", type(ex).__name__, str(ex)[:200])
