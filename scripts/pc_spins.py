"""Tuning: how long the two roles of the producer / consumer renderer wait for each other (build with
BMV_RENDER_DEFS=-DBMV_RENDER_PC_COUNT).  polls / wait: 0 = the flag was ready when the wave got there."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd import _lib
from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch
set_cfg(make_cfg("enerf_eval"))
from boostmvsnerfs_amd.networks.enerf.network import Network
torch.manual_seed(0)
net = Network().eval().cuda()
batch = make_batch(512, 640, device="cuda")
with torch.no_grad():
    for _ in range(3):
        net._forward_checked(dict(batch))
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else _lib.load()
buf = (ctypes.c_ulonglong * 4)()
lib.bmv_debug_fetch_pc_spins.argtypes = [ctypes.c_void_p]
before = list(buf)
lib.bmv_debug_fetch_pc_spins(buf)
a = list(buf)
with torch.no_grad():
    net._forward_checked(dict(batch))
torch.cuda.synchronize()
lib.bmv_debug_fetch_pc_spins(buf)
b = list(buf)
d = [y - x for x, y in zip(a, b)]
print(f"gather waves waiting for an empty mailbox: {d[2]} waits, {d[0] / max(d[2], 1):.1f} polls each")
print(f"MLP waves waiting for a full mailbox:      {d[3]} waits, {d[1] / max(d[3], 1):.1f} polls each")
