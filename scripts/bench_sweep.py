#!/usr/bin/env python3
"""Stand-alone timing of the plane-sweep kernels on the BASELINE config-2 shapes
(HIP events around back-to-back launches; used for kernel iteration and for the
rocprofv3 --pmc passes whose summaries go to profiles/)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd import ops
from boostmvsnerfs_amd.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--level", type=int, default=-1)
ap.add_argument("--narrow", action="store_true", help="level-1 hypotheses in a narrow band (trained-net-like)")
a = ap.parse_args()

dev = "cuda"
H, W = 512, 640
b = make_batch(H, W, device=dev)
torch.manual_seed(0)
levels = {0: dict(C=32, fs=0.25, vs=0.125, D=64), 1: dict(C=16, fs=0.5, vs=0.5, D=8)}
for lvl, L in levels.items():
    if a.level >= 0 and lvl != a.level:
        continue
    Hs, Ws = int(H * L["fs"]), int(W * L["fs"])
    h, w = int(H * L["vs"]), int(W * L["vs"])
    feats = torch.randn(1, 3, L["C"], Hs, Ws, device=dev)
    P = ops.proj_mats(b["src_exts"], b["src_ixts"], b["tar_ext"], b["tar_ixt"], L["fs"], L["vs"])
    if lvl == 0:
        dv, _ = ops.depth_values_uniform(b["near_far"], L["D"], h, w, True)
    else:
        base = 4.0 + 0.5 * torch.rand(1, 1, h, w, device=dev)
        half = 0.15 if a.narrow else 2.0
        dv = (base + torch.linspace(-half, half, L["D"], device=dev).view(1, -1, 1, 1)).contiguous()
    nhwc = ops.nchw_to_nhwc(feats)
    nbytes = 4 * (3 * L["C"] * Hs * Ws + L["C"] * L["D"] * h * w)
    for name, fn in (("nchw direct", lambda: ops.sweep_variance(feats, P, dv, algo=1)),
                     ("cl direct (TA)", lambda: ops.sweep_variance(nhwc, P, dv, channels_last=True, algo=2)),
                     ("cl LDS-staged", lambda: ops.sweep_variance(nhwc, P, dv, channels_last=True, algo=3)),
                     ("transpose", lambda: ops.nchw_to_nhwc(feats))):
        out = torch.empty(1, L["C"], L["D"], h, w, device=dev)
        try:
            for _ in range(5):
                fn()
        except RuntimeError as e:
            print(f"level {lvl} {name:13s}: unsupported ({str(e).split(':')[-1].strip()})")
            continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / a.iters * 1e3
        print(f"level {lvl} {name:13s}: {us:8.2f} us/launch  {nbytes / us / 1e3:8.1f} GB/s algorithmic ({nbytes/1e6:.1f} MB)")
