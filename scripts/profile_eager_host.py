"""Host-side cost of one eager `net(batch)` of the headline workload (what a drop-in run.py calls): cProfile over 100
frames with the GPU kept busy, top functions by own time."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch

set_cfg(make_cfg("enerf_eval"))
torch.manual_seed(0)
from boostmvsnerfs_amd.networks.enerf.network import Network

net = Network().eval().to("cuda")
batch = make_batch(512, 640, device="cuda")
with torch.no_grad():
    for _ in range(5):
        net(dict(batch))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        net(dict(batch))
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
