"""How close to its tolerance is each parameter gradient of the boost fine-tune test, run to run?  (float atomics in
the scatter kernels make the GPU gradients differ by rounding from run to run.)  Prints, per repetition, the worst
err / tol over all tensors and the tensor it belongs to."""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_fixture, tiny_cfg  # noqa: E402


def main():
    from boostmvsnerfs_amd.config import set_cfg
    from boostmvsnerfs_amd.networks.boost_enerf.network import Network
    from boostmvsnerfs_amd.train import NetworkWrapper
    from oracle import enerf as O
    import test_gpu_training as T
    enerf_fx, boost_fx = load_fixture("enerf_tiny"), load_fixture("boost_enerf_tiny")
    tmp = tempfile.mkdtemp()
    cfg = tiny_cfg(boost_fx, "enerf_ours_ft")
    cfg.enerf.cas_config.k_best = len(boost_fx.raw["extra/k_best"])
    cfg.result_dir = tmp
    set_cfg(cfg)
    k_best = [int(k) for k in boost_fx.raw["extra/k_best"]]
    with open(os.path.join(tmp, "view_selection.json"), "w") as f:
        json.dump({"synthetic_0": k_best}, f)
    sd = enerf_fx.group("sd")
    batch = T._targets(boost_fx.batch(), seed=1)
    loss_c, want = T._oracle_grads(lambda s, b, c: O.boost_enerf_forward(s, b, c, k_best), sd, batch, cfg)
    gmax = max(float(g.abs().max()) for g in want.values())
    net = Network()
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda").eval()
    bg = {k: (v.to("cuda") if torch.is_tensor(v) else v) for k, v in batch.items()}
    wrapper = NetworkWrapper(net)
    seen = {}
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        net.zero_grad(set_to_none=True)
        _, loss, _, _ = wrapper(bg)
        loss.backward()
        ratios = {}
        for k, p in net.named_parameters():
            err = (p.grad.cpu() - want[k]).abs()
            tol = 2e-3 * want[k].abs() + 2e-3 * float(want[k].pow(2).mean().sqrt()) + 2e-6 * gmax
            ratios[k] = float((err / tol).max())
            seen[k] = (min(seen.get(k, (9e9, 0))[0], ratios[k]), max(seen.get(k, (9e9, 0))[1], ratios[k]))
        if rep == 0:
            for k in sorted(ratios, key=lambda kk: -ratios[kk])[:3]:
                g, w_ = dict(net.named_parameters())[k].grad.cpu(), want[k]
                err = (g - w_).abs()
                tol = 2e-3 * w_.abs() + 2e-3 * float(w_.pow(2).mean().sqrt()) + 2e-6 * gmax
                i = int((err / tol).argmax())
                print(f"  {k}: element {i} of {w_.numel()}: want {float(w_.flatten()[i]):.4e} got {float(g.flatten()[i]):.4e} "
                      f"rms {float(w_.pow(2).mean().sqrt()):.3e} gmax {gmax:.3e}; entries over half the tolerance: "
                      f"{int((err > 0.5 * tol).sum())}")
        top = sorted(ratios.items(), key=lambda kv: -kv[1])[:3]
        print(f"rep {rep:2d}  " + "  ".join(f"{k} {v:.3f}" for k, v in top), flush=True)
    print("tensors whose worst ratio varies run to run (min .. max):")
    for k, (lo, hi) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
        if hi - lo > 1e-3:
            print(f"  {k:40s} {lo:.3f} .. {hi:.3f}")


if __name__ == "__main__":
    main()
