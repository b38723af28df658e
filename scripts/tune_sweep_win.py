#!/usr/bin/env python3
"""Times the windowed plane sweep (csrc/sweep_win.hip) variants on the sweep inputs of a real frame
(the feature maps, projection matrices and depth hypotheses the ENeRF network hands to the sweep at both
cascade levels of the headline workload), next to the gather kernels, and checks every variant against
the split-geometry kernel.  Kernel-iteration tool; bench.py reports the number that counts."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from boostmvsnerfs_amd import ops
from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--variants", default="")
ap.add_argument("--caps", default="")
ap.add_argument("--H", type=int, default=512)
ap.add_argument("--W", type=int, default=640)
ap.add_argument("--save", default="")
ap.add_argument("--flags", default="0")
ap.add_argument("--ring", default="", help="ring-sweep variants (csrc/sweep_ring.hip, algo 100 + i), e.g. 0,1,2")
ap.add_argument("--zp", default="", help="zero-padded-window variants (csrc/sweep_zp.hip, algo 200 + i; 300 + i on plane-uniform hypotheses)")
ap.add_argument("--ring-env", default="", help="semicolon-separated env settings tried per ring variant, e.g. 'BMV_SWEEP_RING_WPC=2;BMV_SWEEP_RING_WPC=3'")
a = ap.parse_args()

cfg = make_cfg("enerf_eval")
set_cfg(cfg)
torch.manual_seed(0)
from boostmvsnerfs_amd.networks.enerf.network import Network

dev = "cuda"
net = Network().eval().to(dev)
batch = make_batch(a.H, a.W, device=dev)
calls = []


def hook(impl, args, kwargs):
    calls.append((impl, tuple(t.clone() if torch.is_tensor(t) else t for t in args), dict(kwargs)))
    return None


ops.sweep_hook = hook
with torch.no_grad():
    net(batch)
ops.sweep_hook = None
torch.cuda.synchronize()


def timed(fn, iters, per_graph=20):
    """GPU time per launch: `per_graph` launches captured into one HIP graph (issued eagerly from Python a launch
    costs the host ~11 us, more than the kernels under test)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    reps = max(iters // per_graph, 2)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * per_graph) * 1e3


variants = [int(v) for v in a.variants.split(",") if v and int(v) >= 0] if a.variants else list(range(20))
caps = [int(c) for c in a.caps.split(",") if c] or [0]
for lvl, (impl, args, kwargs) in enumerate(calls):
    feats, proj, dv = args[:3]
    cl = feats.permute(0, 1, 3, 4, 2)
    assert cl.is_contiguous(), "expected channel-last features from the conv engine"
    B, S, Hs, Ws, C = cl.shape
    _, D, h, w = dv.shape
    nbytes = 4 * (S * C * Hs * Ws + C * D * h * w)
    print(f"== level {lvl}: C={C} D={D} vol {h}x{w} src {Hs}x{Ws} algorithmic {nbytes/1e6:.1f} MB", flush=True)
    if a.save:
        torch.save({"feats": cl.cpu(), "proj": proj.cpu(), "dv": dv.cpu()}, f"{a.save}_l{lvl}.pt")
    ref = ops._sweep_variance(cl, proj, dv, algo=5, channels_last=True)
    obuf = torch.empty_like(ref)
    scale = ref.abs().mean().item()
    for name, algo in (("split gather", 5), ("tiled gather", 2)):
        us = timed(lambda: ops._sweep_variance(cl, proj, dv, algo=algo, channels_last=True, out=obuf), a.iters)
        print(f"  {name:24s} {us:7.2f} us  {nbytes/us/1e3:7.0f} GB/s  frac {nbytes/us/1e3/8000:.3f}", flush=True)
    for v in variants:
        for fl in [int(f) for f in a.flags.split(",")]:
            os.environ["BMV_SWEEP_WIN_FLAGS"] = str(fl)
            for cap in caps:
                if cap:
                    os.environ["BMV_SWEEP_WIN_CAP"] = str(cap)
                else:
                    os.environ.pop("BMV_SWEEP_WIN_CAP", None)
                try:
                    out = ops._sweep_variance(cl, proj, dv, algo=40 + v, channels_last=True)
                    torch.cuda.synchronize()
                except RuntimeError as e:
                    print(f"  variant {v:2d} cap {cap:4d}: {str(e)[:90]}")
                    continue
                err = (out - ref).abs().max().item()
                us = timed(lambda: ops._sweep_variance(cl, proj, dv, algo=40 + v, channels_last=True, out=obuf), a.iters)
                print(f"  variant {v:2d} cap {cap:4d} flags {fl}  {us:7.2f} us  {nbytes/us/1e3:7.0f} GB/s  "
                      f"frac {nbytes/us/1e3/8000:.3f}  max|d| {err:.2e} (mean|ref| {scale:.2e})", flush=True)
    pu = bool((dv == dv[:, :, :1, :1]).all())
    for v in [int(x) for x in a.zp.split(",") if x]:
        for base in ([200, 300] if pu else [200]):
            try:
                out = ops._sweep_variance(cl, proj, dv, algo=base + v, channels_last=True)
                torch.cuda.synchronize()
                err = (out - ref).abs().max().item()
                us = timed(lambda: ops._sweep_variance(cl, proj, dv, algo=base + v, channels_last=True, out=obuf), a.iters)
                print(f"  zp {v:2d} {'plane-uniform' if base == 300 else 'per-voxel   '}  {us:7.2f} us  {nbytes/us/1e3:7.0f} GB/s  "
                      f"frac {nbytes/us/1e3/8000:.3f}  max|d| {err:.2e} (mean|ref| {scale:.2e})", flush=True)
            except RuntimeError as e:
                print(f"  zp {v:2d}: {str(e)[:90]}")
    for v in [int(x) for x in a.ring.split(",") if x]:
        for envs in (a.ring_env.split(";") if a.ring_env else [""]):
            sets = dict(kv.split("=") for kv in envs.split(",") if kv)
            os.environ.update(sets)
            try:
                out = ops._sweep_variance(cl, proj, dv, algo=100 + v, channels_last=True)
                torch.cuda.synchronize()
                err = (out - ref).abs().max().item()
                us = timed(lambda: ops._sweep_variance(cl, proj, dv, algo=100 + v, channels_last=True, out=obuf), a.iters)
                print(f"  ring {v:2d} {envs:40s} {us:7.2f} us  {nbytes/us/1e3:7.0f} GB/s  frac {nbytes/us/1e3/8000:.3f}  "
                      f"max|d| {err:.2e} (mean|ref| {scale:.2e})", flush=True)
            except RuntimeError as e:
                print(f"  ring {v:2d} {envs}: {str(e)[:90]}")
            finally:
                for k in sets:
                    os.environ.pop(k, None)
os.environ.pop("BMV_SWEEP_WIN_CAP", None)
os.environ.pop("BMV_SWEEP_WIN_FLAGS", None)
