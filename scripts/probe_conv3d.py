"""Probe: MIOpen conv3d forward/backward cost of the ENeRF level-1 regulariser in NCDHW vs channels_last_3d."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd.config import make_cfg, set_cfg
set_cfg(make_cfg("enerf_eval"))
from boostmvsnerfs_amd.networks.enerf.cnn import CostRegNet
torch.manual_seed(0)
for fmt in ("contiguous", "channels_last_3d"):
    net = CostRegNet(16).cuda().train()
    x = torch.randn(1, 16, 8, 256, 320, device="cuda")
    if fmt == "channels_last_3d":
        net = net.to(memory_format=torch.channels_last_3d)
        x = x.to(memory_format=torch.channels_last_3d)
    x.requires_grad_(True)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f, p = net(x)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (f.sum() + p.sum()).backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{fmt:18s} iter {it}: fwd {1e3*(t1-t0):8.2f} ms  bwd {1e3*(t2-t1):8.2f} ms", flush=True)
