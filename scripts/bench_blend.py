#!/usr/bin/env python3
"""a16 fusion kernel (bmv_blend_fwd) at the K-volume workloads' own shapes, on the stacks their frames produce; and the
share of (volume, sample) MLP evaluations whose normalised mask is exactly 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd import ops

dev = "cuda"
for name, (K, N, Ns) in {"config 4 (mvsnerf_ours 224x352, 128 samples)": (4, 224 * 352, 128),
                         "config 3 (enerf_ours 480x736, 2 samples)": (4, 480 * 736, 2)}.items():
    g = torch.Generator().manual_seed(0)
    raws = torch.rand(1, K, N, Ns, 4, generator=g).to(dev)
    masks = (torch.randint(0, 4, (1, K, N, Ns), generator=g).float() / 3).to(dev)
    z = (torch.rand(1, K, N, Ns, generator=g) + 2).to(dev)
    for _ in range(3):
        ops.blend(raws, masks, z, normalise=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.blend(raws, masks, z, normalise=True)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    nbytes = K * N * Ns * 24 + N * (Ns + 4) * 4
    print(f"{name}: {us:8.1f} us  {nbytes / 1e6:7.1f} MB  {nbytes / us / 1e3:7.1f} GB/s = {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s")
