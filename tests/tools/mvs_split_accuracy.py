"""Accuracy of MVSNeRF's fused 6 x 128 MLP (Renderer_ours, lib/networks/mvsnerf/network.py:201-229) in its two matrix
forms against a float64 evaluation of the same network on the same fp32 inputs and weights:
(i) bmv_tuning BMV_MVS_SPLIT=0: every layer on fp32 MFMAs, (ii) the default 1: the ten 128 -> 128 weight chunks
(pts_linears.1-4, feature_linear) as bf16 MFMAs on three-piece fp32 operands.  Printed per trial (the fixture's weights;
3 x larger weight matrices): max / mean |error| of rgb and alpha for torch's CPU fp32 (the oracle), for (i) and for (ii),
and the distance between the two HIP forms.

    python tests/tools/mvs_split_accuracy.py > profiles/r6/mvs_split_accuracy.txt
"""
import os
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from boostmvsnerfs_amd import _lib, ops  # noqa: E402
from conftest import load_fixture  # noqa: E402
from oracle import mvsnerf as M  # noqa: E402  (the checker: test infrastructure)


def main():
    fx = load_fixture("mvsnerf_tiny")
    keys = sorted((k for k in fx.raw if k.startswith("cap/run_network_mvs#")), key=lambda k: int(k.split("#")[1].split(".")[0]))
    x0 = torch.cat([fx.t(k) for k in keys], 0).reshape(-1, 86)
    P = 1 << 18
    for trial, wscale in enumerate((1.0, 3.0)):
        torch.manual_seed(5 + trial)
        sd = {k: v.clone() for k, v in fx.group("sd").items() if k.startswith("nerf.nerf.")}
        for k in sd:
            sd[k] = sd[k] * wscale if k.endswith(".weight") else sd[k] + 0.05 * torch.randn_like(sd[k])
        x = x0[torch.randint(0, x0.shape[0], (P,))].contiguous()
        x[:, 63:83] += 0.05 * torch.randn(P, 20)
        want = M.renderer_mlp({k: v.double() for k, v in sd.items()}, x.double())
        ref32 = M.renderer_mlp(sd, x)
        names = ops.MVS_MLP_PARAM_ORDER
        blob = ops.mvs_mlp_pack_weights({k: sd[f"nerf.nerf.{k}.weight"].cuda() for k in names},
                                        {k: sd[f"nerf.nerf.{k}.bias"].cuda() for k in names})
        outs = {}
        for split in (0, 1):
            _lib.set_tuning("BMV_MVS_SPLIT", split)
            outs[split] = ops.mvs_mlp(x.cuda(), blob).cpu()
        _lib.set_tuning("BMV_MVS_SPLIT", None)
        print(f"trial {trial} (weight matrices x {wscale}): {P} points; alpha up to {float(want[..., 3].abs().max()):.2f}")
        for name, got in (("torch CPU fp32 (the oracle)", ref32), ("HIP, fp32 MFMAs (BMV_MVS_SPLIT=0)", outs[0]),
                          ("HIP, 128 -> 128 chunks bf16 x 3 (default)", outs[1])):
            e = (got.double() - want).abs()
            print(f"    {name:44s} rgb max {float(e[..., :3].max()):.3e} mean {float(e[..., :3].mean()):.3e}   "
                  f"alpha max {float(e[..., 3].max()):.3e} mean {float(e[..., 3].mean()):.3e}")
        d = (outs[1] - outs[0]).abs()
        print(f"    split against fp32 MFMAs: max {float(d.max()):.3e}, {float((d > 0).float().mean()) * 100:.1f} % of the values differ")


if __name__ == "__main__":
    main()
