#!/usr/bin/env python3
"""Share of (volume, sample) MLP evaluations of the K-volume frames whose fused weight is exactly zero: normalised mask
m_k / sum_k m_k = 0 while another volume sees the sample (enerf/utils.py:639-667, boost_enerf/network.py:163-170), per
sample and per 32-sample MLP tile (a tile can be skipped only if ALL its samples are such)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from boostmvsnerfs_amd import ops

for wl in ("mvsnerf_ours_224x352_128planes_k4", "enerf_ours_480x736_6src_k4"):
    args = argparse.Namespace(workload=wl, shard="views", sweep_algo=0, stub_renderer=False)
    cfg, w, net, sd, batch_cpu, batch, level = bench.build(args, 0, torch.device("cuda", 0))
    seen = []
    orig = ops.blend
    def grab(raws, masks, z, normalise=True, **kw):
        seen.append(masks.detach().clone())
        return orig(raws, masks, z, normalise=normalise, **kw)
    ops.blend = grab
    for mod in list(sys.modules.values()):
        if getattr(mod, "ops", None) is ops:
            pass
    with torch.no_grad():
        fwd = getattr(net, "_forward_checked", net)
        fwd(dict(batch))
    ops.blend = orig
    torch.cuda.synchronize()
    for m in seen:
        B, K, N, Ns = m.shape
        msum = m.sum(1, keepdim=True)
        dead = (m == 0) & (msum > 0)                       # evaluated, multiplied by exactly 0
        tile = 32 if Ns >= 32 else Ns
        dt = dead.reshape(B, K, -1, tile).all(-1) if (N * Ns) % tile == 0 else None
        print(f"{wl}: masks {tuple(m.shape)}  dead samples {float(dead.float().mean()):.3f}  "
              f"no volume sees the sample {float((msum == 0).float().mean()):.3f}  "
              + (f"dead {tile}-sample tiles {float(dt.float().mean()):.3f}" if dt is not None else ""))
