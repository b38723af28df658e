"""Phase stamps of the fused ENeRF renderer: shader-clock cycles per 32-sample tile and wave (third tile of every wave).
Needs a tuning build of render.hip with -DBMV_RENDER_STAMPS (hipcc ... -DBMV_RENDER_STAMPS -c render.hip, relink libbmv.so);
the stamps force `s_waitcnt` at every phase boundary, so they serialise what the shipped kernel overlaps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch
cfg = make_cfg("enerf_eval"); set_cfg(cfg)
torch.manual_seed(0)
from boostmvsnerfs_amd.networks.enerf.network import Network
net = Network().eval().to("cuda")
batch = make_batch(512, 640, device="cuda")
with torch.no_grad():
    for _ in range(2):
        out = net(batch)
torch.cuda.synchronize()
import ctypes
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "boostmvsnerfs_amd", "libbmv.so"))
buf = (ctypes.c_float * (512 * 4 * 8))()
assert lib.bmv_debug_fetch_stamps(buf) == 0
w = np.frombuffer(buf, dtype=np.float32).reshape(-1, 8)
w = w[w[:, 0] == 1.0]
names = ["geometry (rays, bounds, sample)", "volume taps (trilinear)", "image taps x3 + dir", "MLP"]
prev = 0
print(len(w), "waves")
for i, n in enumerate(names):
    cur = w[:, i + 1]
    print(f"  {n:34s} +{np.median(cur - prev):8.0f} cyc   (cum {np.median(cur):8.0f})")
    prev = cur
print("  tile period (start to start)     ", np.median(w[:, 5]))
