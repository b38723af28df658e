#!/usr/bin/env python3
"""fp64 arbitration of the K-volume fine-tune gradients.

test_boost_enerf_finetune_gradients holds the HIP gradients to the fp32 ORACLE's; its worst entry sat at 0.96 x the
tolerance.  Is that an error of the HIP path or the conditioning of the fp32 reference itself?  This tool computes the
oracle's gradients in fp32 and in fp64 (same graph, torch.set_default_dtype) and -- with --gpu -- the HIP path's, and
prints per tensor max |x - fp64| / tol for x = fp32 oracle and x = HIP.  If the HIP path is no further from the fp64 truth
than the fp32 oracle is, the margin is conditioning.

    python tests/tools/grad_fp64_arbitration.py [--gpu]
"""
import json
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_fixture, tiny_cfg   # noqa: E402
from boostmvsnerfs_amd.config import set_cfg   # noqa: E402
from oracle import enerf as O                  # noqa: E402

enerf_fx, boost_fx = load_fixture("enerf_tiny"), load_fixture("boost_enerf_tiny")
cfg = tiny_cfg(boost_fx, "enerf_ours_ft")
k_best = [int(k) for k in boost_fx.raw["extra/k_best"]]
cfg.enerf.cas_config.k_best = len(k_best)
tmp = tempfile.mkdtemp()
cfg.result_dir = tmp
set_cfg(cfg)
with open(os.path.join(tmp, "view_selection.json"), "w") as f:
    json.dump({"synthetic_0": k_best}, f)
sd = enerf_fx.group("sd")
b = boost_fx.batch()
g = torch.Generator().manual_seed(1)
for i in range(2):
    b[f"rgb_{i}"] = torch.rand(1, b[f"rays_{i}"].shape[1], 3, generator=g)
cc = cfg.enerf.cas_config


def oracle_grads(dtype):
    torch.set_default_dtype(dtype)
    try:
        leaves = {k: (v.detach().to(dtype) if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and "running" not in k)
                  for k, v in sd.items()}
        bb = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        out = O.boost_enerf_forward(leaves, bb, cfg, k_best)
        loss = sum(cc.loss_weight[i] * ((out[f"rgb_level{i}"] - bb[f"rgb_{i}"]) ** 2).mean() for i in range(cc.num) if f"rgb_level{i}" in out)
        loss.backward()
        return float(loss.detach()), {k: v.grad.double() for k, v in leaves.items() if v.requires_grad}
    finally:
        torch.set_default_dtype(torch.float32)


l32, g32 = oracle_grads(torch.float32)
l64, g64 = oracle_grads(torch.float64)
print(f"loss fp32 {l32:.9f}  fp64 {l64:.9f}")
ghip = None
if "--gpu" in sys.argv:
    from boostmvsnerfs_amd.networks.boost_enerf.network import Network
    from boostmvsnerfs_amd.train import NetworkWrapper
    net = Network()
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    bg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    _, loss, _, _ = NetworkWrapper(net)(bg)
    loss.backward()
    ghip = {k: p.grad.double().cpu() for k, p in net.named_parameters()}
    print(f"loss HIP  {float(loss):.9f}")
gmax = max(float(v.abs().max()) for v in g64.values())
rows = []
for k, w in g64.items():
    tol = 2e-3 * w.abs() + 2e-3 * float(w.pow(2).mean().sqrt()) + 2e-6 * gmax
    r32 = float(((g32[k] - w).abs() / tol).max())
    rh = float(((ghip[k] - w).abs() / tol).max()) if ghip else float("nan")
    rhc = float(((ghip[k] - g32[k]).abs() / (2e-3 * g32[k].abs() + 2e-3 * float(g32[k].pow(2).mean().sqrt()) + 2e-6 * gmax)).max()) if ghip else float("nan")
    rows.append((max(r32, rh if rh == rh else 0.0), k, r32, rh, rhc))
rows.sort(reverse=True)
print("worst tensors: max |x - fp64| / tolerance   (x = fp32 oracle | HIP)   and HIP vs fp32 oracle (the test's comparison)")
for _, k, r32, rh, rhc in rows[:10]:
    print(f"  {k:42s} fp32 oracle {r32:6.3f}   HIP {rh:6.3f}   HIP vs fp32 oracle {rhc:6.3f}")
print(f"over all {len(rows)} tensors: fp32 oracle max {max(r[2] for r in rows):.3f}" + (f", HIP max {max(r[3] for r in rows):.3f}, HIP vs fp32 oracle max {max(r[4] for r in rows):.3f}" if ghip else ""))
