"""Time the ENeRF MLP forward and backward (autograd.NerfMLP) alone at the fine-tune workload's sizes
(512x640: level 1 = 327680 rays x 2 samples, feat_ch 8; level 0 = 20480 rays x 8 samples, feat_ch 32)."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import autograd as A, ops  # noqa: E402


def params_for(feat_ch, dev):
    FC = feat_ch + 3
    shapes = {"agg.view_fc.0": (FC, 4), "agg.global_fc.0": (32, FC * 3), "agg.agg_w_fc.0": (1, 32), "agg.fc.0": (16, 32),
              "lr0.0": (64, 24), "sigma.0": (1, 64), "color.0": (64, 88 + FC + 4), "color.2": (1, 64)}
    out = []
    for n in ops.NERF_PARAM_ORDER:
        o, i = shapes[n]
        out.append((torch.randn(o, i, device=dev) / i ** 0.5).requires_grad_(True))
        out.append((torch.randn(o, device=dev) * 0.1).requires_grad_(True))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    for feat_ch, P in ((8, 327680 * 2), (32, 20480 * 8)):
        FC = feat_ch + 3
        params = params_for(feat_ch, dev)
        vox = torch.randn(1, P, 8, device=dev, requires_grad=True)
        img = torch.randn(1, P, 3, FC + 4, device=dev, requires_grad=True)
        g = torch.randn(1, P, 4, device=dev)
        for phase in ("fwd", "fwd+bwd"):
            def step():
                out = A.NerfMLP.apply(vox, img, feat_ch, *params)
                if phase != "fwd":
                    out.backward(g)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                step()
            e1.record()
            torch.cuda.synchronize()
            print(f"feat_ch {feat_ch:2d}  P {P:7d}  {phase:8s} {e0.elapsed_time(e1) / a.iters:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
