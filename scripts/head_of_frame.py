"""The first kernels of a frame under rocprofv3: fresh tensors every call (feed ring node first) against a declared
resident batch (no feed node).  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -- python scripts/head_of_frame.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd.config import make_cfg, set_cfg  # noqa: E402
from boostmvsnerfs_amd.networks.enerf.network import Network  # noqa: E402
from boostmvsnerfs_amd.synthetic import clone_batch, make_batch  # noqa: E402

cfg = make_cfg("enerf_eval")
set_cfg(cfg)
torch.manual_seed(0)
net = Network().eval().cuda()
batches = [clone_batch(make_batch(512, 640, seed=s), "cuda") for s in range(3)]
with torch.no_grad():
    for i in range(30):
        net(batches[i % 3])
        torch.cuda.synchronize()
    net.resident_inputs = net.alias_outputs = True
    for i in range(30):
        net(batches[0])
        torch.cuda.synchronize()
print("done")
