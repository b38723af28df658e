#!/usr/bin/env python3
"""Launches each sweep kernel a few times on the frame's sweep inputs (for rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd import ops
from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch
algos = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,40,49").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = make_cfg("enerf_eval"); set_cfg(cfg)
torch.manual_seed(0)
from boostmvsnerfs_amd.networks.enerf.network import Network
net = Network().eval().to("cuda")
batch = make_batch(512, 640, device="cuda")
calls = []
ops.sweep_hook = lambda impl, args, kwargs: (calls.append(tuple(t.clone() for t in args[:3])), None)[1]
with torch.no_grad():
    net(batch)
ops.sweep_hook = None
torch.cuda.synchronize()
for feats, proj, dv in calls:
    cl = feats.permute(0, 1, 3, 4, 2)
    for algo in algos:
        for _ in range(reps):
            ops._sweep_variance(cl, proj, dv, algo=algo, channels_last=True)
        torch.cuda.synchronize()
