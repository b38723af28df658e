"""Accuracy of the fused MLP's two forms against a float64 evaluation of the same network on the same fp32 inputs:
(i) every chain on fp32 MFMAs (the default), (ii) bmv_tuning BMV_RENDER_SPLIT=1: color.0's shared part on the bf16 matrix
pipe with both operands split into three bf16 pieces.  Printed: max / mean |error| of rgb and sigma against the float64
result, for each form, and the distance between the two forms.

    python tests/tools/mlp_split_accuracy.py > profiles/r5/mlp_split_accuracy.txt
"""
import sys

import torch

sys.path.insert(0, ".")
from boostmvsnerfs_amd import _lib, ops  # noqa: E402
from boostmvsnerfs_amd.config import make_cfg, set_cfg  # noqa: E402
from oracle import enerf as O  # noqa: E402  (the checker: test infrastructure)


def main():
    set_cfg(make_cfg("enerf_eval"))
    from boostmvsnerfs_amd.networks.enerf.nerf import NeRF
    torch.manual_seed(7)
    P = 1 << 18
    for trial, wscale in enumerate((1.0, 3.0)):
        net = NeRF(feat_ch=8 + 3)
        with torch.no_grad():
            for p_ in net.parameters():           # non-zero biases, wider pre-activations in the second trial
                if p_.dim() == 1:
                    p_.normal_(0, 0.1)
                else:
                    p_.mul_(wscale)
        sd = {("nerf." + k): v.detach() for k, v in net.state_dict().items()}
        vox = torch.randn(1, P, 8)
        img = torch.cat([torch.randn(1, P, 3, 8), torch.rand(1, P, 3, 3), torch.randn(1, P, 3, 4) * 0.5], -1)
        want = O.nerf_mlp({k: v.double() for k, v in sd.items()}, "nerf.", vox.double(), img.double())
        ref32 = O.nerf_mlp(sd, "nerf.", vox, img)
        netd = net.to("cuda").eval()
        outs = {}
        with torch.no_grad():
            for split in (0, 1):
                _lib.set_tuning("BMV_RENDER_SPLIT", split)
                outs[split] = ops.nerf_mlp(vox.cuda(), img.cuda(), netd.packed_weights(), 8).cpu()
        _lib.set_tuning("BMV_RENDER_SPLIT", None)
        print(f"trial {trial} (weights x {wscale}): {P} samples, 3 views; |sigma| up to {float(want[..., 3].abs().max()):.2f}")
        for name, got in (("torch CPU fp32 (the oracle)", ref32), ("HIP, fp32 MFMAs (default)", outs[0]), ("HIP, two-tile chains split bf16 x 3", outs[1])):
            e = (got.double() - want).abs()
            print(f"    {name:42s} rgb max {float(e[..., :3].max()):.3e} mean {float(e[..., :3].mean()):.3e}   "
                  f"sigma max {float(e[..., 3].max()):.3e} mean {float(e[..., 3].mean()):.3e}")
        d = (outs[1] - outs[0]).abs()
        print(f"    split against default: max {float(d.max()):.3e}, {float((d > 0).float().mean()) * 100:.1f} % of the values differ")


if __name__ == "__main__":
    main()
