#!/usr/bin/env python3
"""The quad-planar planned sweep (csrc/sweep_quad.hip) on the FRAME'S OWN sweep inputs (512x640, both cascade levels):
parity against the windowed channel-last kernel (algo 4) and dispatch-bound kernel times of the tuning variants.

    python scripts/bench_sweep_quad.py [--variants 0,3,7 ...] [--flags 0] [--cold]

--cold: a 512 MB fill between timed launches (the source maps then come from HBM / cold L2, as inside a frame).
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd import ktimer, ops
from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--cold", action="store_true")
ap.add_argument("--evict-mb", type=int, default=0, help="MB written between timed launches (32: the L2s turn over, the "
                "Infinity Cache keeps the source maps; 512 = --cold: everything comes from HBM)")
ap.add_argument("--size", default="512x640")
a = ap.parse_args()
H, W = (int(v) for v in a.size.split("x"))

cfg = make_cfg("enerf_eval")
set_cfg(cfg)
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
mb = 512 if a.cold else a.evict_mb
scratch = torch.empty(mb << 18, device="cuda") if mb else None


def timed(fn, name):
    ktimer.reset()
    ktimer.enabled, ktimer.only = True, ("sweep_variance",)
    for _ in range(a.iters):
        if scratch is not None:
            scratch.fill_(1.0)
        fn()
    torch.cuda.synchronize()
    ktimer.collect()
    ks = ktimer.summary()
    ktimer.enabled = False
    return {k: (v[1] * 1e3, v[2] * 1e3) for k, v in ks.items()}


for lvl, (feats, proj, dv) in enumerate(calls):
    if feats.dim() == 6:      # already quad-planar (the network's own layout): back to channel-last for the reference kernel
        quad = feats
        B, V, Q, Hs, Ws, _ = quad.shape
        cl = quad.permute(0, 1, 3, 4, 2, 5).reshape(B, V, Hs, Ws, Q * 4).contiguous()
    else:
        cl = feats.permute(0, 1, 3, 4, 2).contiguous()
        quad = ops.to_quad_planar(cl, channels_last=True)
    B, S, Hs, Ws, C = cl.shape
    _, D, h, w = dv.shape
    nbytes = 4 * (S * C * Hs * Ws + C * D * h * w)
    pu = bool((dv[:, :, :1, :1] == dv).all())
    want = ops._sweep_variance(cl, proj, dv, algo=4, channels_last=True)
    t = timed(lambda: ops._sweep_variance(cl, proj, dv, algo=4, channels_last=True), "win")
    for k, (avg, mn) in t.items():
        print(f"level {lvl}  windowed (algo 4)        {k}: avg {avg:6.2f} us  min {mn:6.2f}  -> {nbytes / avg / 1e3 / 8000:.3f} of 8 TB/s")
    variants = [int(v) for v in a.variants.split(",")] if a.variants else ([1, 4, 10, 12, 13, 14, 15] if Ws > 1.5 * w else [0, 3, 5, 11, 16])
    for v in variants:
        try:
            got = ops._sweep_variance_quad(quad, proj, dv, plane_uniform=pu, variant=v)
        except RuntimeError as e:
            print(f"level {lvl}  quad variant {v}: {e}")
            continue
        torch.cuda.synchronize()
        rms = float(want.pow(2).mean().sqrt())
        err = float(((got - want).abs() / (want.abs() + rms)).max())
        t = timed(lambda: ops._sweep_variance_quad(quad, proj, dv, plane_uniform=pu, variant=v, flags=a.flags), "quad")
        for k, (avg, mn) in t.items():
            if k.startswith("sweep_variance"):
                print(f"level {lvl}  quad variant {v} (pu={int(pu)}) {k}: avg {avg:6.2f} us  min {mn:6.2f}  -> {nbytes / avg / 1e3 / 8000:.3f} of 8 TB/s"
                      f"   max rel err vs windowed {err:.2e}")
