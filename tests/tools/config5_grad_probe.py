"""Per-tensor distance of the HIP parameter gradients from the oracle's for the config-5 step at FULL size (480x736,
N = 6, K = 4, both levels; rays ::STRIDE) -- tests/test_gpu_fullsize.py::test_config5_full_size_gradients... prints only
its first failure.  python tests/tools/config5_grad_probe.py [STRIDE] [--level0-only|--level1-only]"""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_fullsize as T
    from boostmvsnerfs_amd import _lib
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.boost_enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from boostmvsnerfs_amd.train import NetworkWrapper
    from oracle import enerf as O
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    stride = int(args[0]) if args else 16
    H, W = (480, 736) if "--small" not in sys.argv else (96, 160)
    cfg = make_cfg("enerf_ours_ft")
    cc = cfg.enerf.cas_config
    cc.volume_planes = [64, 8]
    cc.k_best = 4
    if "--level1-only" in sys.argv:
        cc.render_if = [False, True]
    cfg.result_dir = tempfile.mkdtemp()
    set_cfg(cfg)
    sel = [0, 7, 13, 19]
    with open(os.path.join(cfg.result_dir, "view_selection.json"), "w") as f:
        json.dump({"synthetic_0": sel}, f)
    torch.manual_seed(0)
    net = Network().eval()
    if "--no-perturb" not in sys.argv:
        net = T._perturb(net)
    batch = make_batch(H, W, n_views=6, seed=0)
    g = torch.Generator().manual_seed(3)
    for i in range(cc.num):
        batch[f"rays_{i}"] = batch[f"rays_{i}"][:, ::stride].contiguous()
        batch[f"rgb_{i}"] = torch.rand(1, batch[f"rays_{i}"].shape[1], 3, generator=g)
    def oracle(dtype):
        torch.set_default_dtype(dtype)
        try:
            leaves = {k: (v.detach().to(dtype) if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and "running" not in k)
                      for k, v in net.state_dict().items()}
            bb = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in clone_batch(batch).items()}
            out = O.boost_enerf_forward(leaves, bb, cfg, sel)
            loss = sum(cc.loss_weight[i] * ((out[f"rgb_level{i}"] - bb[f"rgb_{i}"]) ** 2).mean()
                       for i in range(cc.num) if f"rgb_level{i}" in out)
            loss.backward()
            return loss.detach().float(), {k: v.grad.float() for k, v in leaves.items() if v.requires_grad and v.grad is not None}
        finally:
            torch.set_default_dtype(torch.float32)
    loss_c, want = oracle(torch.float32)
    want64 = None
    if "--fp64" in sys.argv:       # the oracle in float64 = the truth both fp32 results are measured against
        loss64, want64 = oracle(torch.float64)
        print(f"oracle loss fp32 {float(loss_c):.9f}  fp64 {float(loss64):.9f}")
    net = net.to("cuda")
    bg = clone_batch(batch, "cuda")
    for det in (0, 1):
        _lib.set_tuning("BMV_DETERMINISTIC", det)
        net.zero_grad(set_to_none=True)
        _, loss, _, _ = NetworkWrapper(net)(bg)
        loss.mean().backward()
        gmax = max(float(v.abs().max()) for v in want.values())
        rows = []
        for k, p in net.named_parameters():
            if k not in want or p.grad is None:
                rows.append((k, float("nan"), 0, 0))
                continue
            w_ = want[k]
            d = (p.grad.cpu() - w_)
            rel = float(d.pow(2).sum().sqrt()) / (float(w_.pow(2).sum().sqrt()) + 1e-6 * gmax)
            tol = 2e-3 * w_.abs() + 2e-3 * float(w_.pow(2).mean().sqrt()) + 2e-6 * gmax
            rows.append((k, rel, float((d.abs() / tol).max()), float(w_.abs().max())))
        print(f"deterministic={det}  loss HIP {float(loss):.8f} oracle {float(loss_c):.8f}  gmax {gmax:.3e}")
        for k, rel, ratio, mx in sorted(rows, key=lambda r: -r[1])[:25]:
            print(f"   {k:44s} rel L2 {rel:.3e}   worst entry / tol {ratio:7.2f}   |want|max {mx:.3e}")
        print("   ... tensors over 1e-2:", sum(r[1] > 1e-2 for r in rows), " over 1e-3:", sum(r[1] > 1e-3 for r in rows), "of", len(rows))
        if want64 is not None:
            print("   against the float64 oracle: worst entry / tol of   HIP | fp32 oracle   (tol = the test's 2e-3 bar on the fp64 gradient)")
            arb = []
            for k, p in net.named_parameters():
                w64 = want64[k]
                tol = 2e-3 * w64.abs() + 2e-3 * float(w64.pow(2).mean().sqrt()) + 2e-6 * gmax
                rh = float(((p.grad.cpu() - w64).abs() / tol).max())
                r32 = float(((want[k] - w64).abs() / tol).max())
                l2h = float((p.grad.cpu() - w64).pow(2).sum().sqrt()) / (float(w64.pow(2).sum().sqrt()) + 1e-6 * gmax)
                l232 = float((want[k] - w64).pow(2).sum().sqrt()) / (float(w64.pow(2).sum().sqrt()) + 1e-6 * gmax)
                arb.append((max(rh, r32), k, rh, r32, l2h, l232))
            for _, k, rh, r32, l2h, l232 in sorted(arb, reverse=True)[:20]:
                print(f"   {k:44s} HIP {rh:6.2f} | fp32 oracle {r32:6.2f}     rel L2: HIP {l2h:.2e} | fp32 oracle {l232:.2e}")
            print(f"   worst over all tensors: HIP {max(a[2] for a in arb):.2f} | fp32 oracle {max(a[3] for a in arb):.2f};"
                  f"  tensors where HIP is the farther one: {sum(a[2] > a[3] for a in arb)} of {len(arb)}")


if __name__ == "__main__":
    main()
