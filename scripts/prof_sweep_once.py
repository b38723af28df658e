#!/usr/bin/env python3
"""Launches the plane sweeps a few times on the FRAME'S OWN sweep inputs (feature maps, projection matrices and
hypotheses captured from one forward of the workload), for rocprofv3 --pmc / --kernel-trace passes.

    python scripts/prof_sweep_once.py [algos, default "0,4"] [reps, default 3] [HxW, default 512x640]

algo 0 = the shipped kernels (quad-planar source maps, level defaults, quad-record output), 1 = the same with the planar
output, 4 = the windowed channel-last kernel,
500 + i / 600 + i = tuning variants of the quad kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd import ops
from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch
algos = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "512x640").split("x"))
set_cfg(make_cfg("enerf_eval"))
torch.manual_seed(0)
from boostmvsnerfs_amd.networks.enerf.network import Network
net = Network().eval().to("cuda")
batch = make_batch(H, W, device="cuda")
calls = []
ops.sweep_hook = lambda impl, args, kwargs: (calls.append(tuple((t.data if isinstance(t, ops.QuadFeats) else t).clone() for t in args[:3])), None)[1]
with torch.no_grad():
    net._forward_checked(dict(batch))
ops.sweep_hook = None
torch.cuda.synchronize()
for feats, proj, dv in calls:
    if feats.dim() == 6:
        quad = feats
        B, V, Q, Hs, Ws, _ = quad.shape
        cl = quad.permute(0, 1, 3, 4, 2, 5).reshape(B, V, Hs, Ws, Q * 4).contiguous()
    else:
        cl = feats.permute(0, 1, 3, 4, 2).contiguous()
        quad = ops.to_quad_planar(cl, channels_last=True)
    pu = bool((dv[:, :, :1, :1] == dv).all())
    for algo in algos:
        for _ in range(reps):
            if algo == 0:      # the frame's form: the variance leaves as quad records (round 5)
                ops._sweep_variance_quad(quad, proj, dv, plane_uniform=pu, variant=-1, quad_out=True)
            elif algo == 1:    # ... and with the planar output
                ops._sweep_variance_quad(quad, proj, dv, plane_uniform=pu, variant=-1)
            elif algo >= 500:
                ops._sweep_variance_quad(quad, proj, dv, plane_uniform=algo >= 600 and pu, variant=algo % 100)
            else:
                ops._sweep_variance(cl, proj, dv, algo=algo, channels_last=True)
        torch.cuda.synchronize()
