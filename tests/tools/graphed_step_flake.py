"""Is the graphed-vs-eager gradient mismatch of tests/test_gpu_training.py::test_graphed_train_step_equals_eager_steps
noise of the float atomics (amplified by training-mode batch norm over the few voxels of the tiny fixture's deep
levels) or a replay that reads a stale / un-zeroed buffer?  

Runs the test's lockstep REPS times with THREE twins: A = GraphedTrainStep (3 eager steps, capture + first replay at step
3, pure replays at steps 4 and 5), B and C = eager `train_step`; before every step B and C take over A's parameters,
batch-norm statistics and Adam state.  Logs, per step, the relative L2 distance of every parameter gradient for the pairs
A-B (graphed vs eager) and B-C (eager vs eager), and prints the distribution per step kind at the end.  If the replay
steps sat above the eager-vs-eager distribution, the captured backward would read something stale.

    python tests/tools/graphed_step_flake.py [REPS] [--det]      (--det: bmv_tuning BMV_DETERMINISTIC = 1)
"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_fixture, tiny_cfg  # noqa: E402

DEV = "cuda"


def rel_l2(ga, gb, gmax):
    num = float((ga - gb).pow(2).sum().sqrt())
    den = float(gb.pow(2).sum().sqrt()) + 1e-6 * gmax
    return num / den


def one_rep(enerf_fx, log):
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.train import GraphedTrainStep, NetworkWrapper, make_optimizer, train_step
    net_a = Network()
    net_a.load_state_dict(enerf_fx.group("sd"), strict=True)
    net_a = net_a.to(DEV).train()
    net_b, net_c = copy.deepcopy(net_a).train(), copy.deepcopy(net_a).train()
    base = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in enerf_fx.batch().items()}
    wa, wb, wc = NetworkWrapper(net_a), NetworkWrapper(net_b), NetworkWrapper(net_c)
    oa, ob, oc = make_optimizer(net_a), make_optimizer(net_b), make_optimizer(net_c)
    graphed = GraphedTrainStep(wa, oa)
    for s in range(6):
        b = dict(base)
        g = torch.Generator().manual_seed(s)
        for i in range(2):
            b[f"rgb_{i}"] = torch.rand(1, b[f"rays_{i}"].shape[1], 3, generator=g).to(DEV)
        for net, opt in ((net_b, ob), (net_c, oc)):
            net.load_state_dict(copy.deepcopy(net_a.state_dict()))
            opt.load_state_dict(copy.deepcopy(oa.state_dict()))
        loss_a, _ = graphed(dict(b))
        loss_b, _ = train_step(wb, ob, dict(b))
        loss_c, _ = train_step(wc, oc, dict(b))
        ga = {k: p.grad for k, p in net_a.named_parameters()}
        gc_ = {k: p.grad for k, p in net_c.named_parameters()}
        gmax = max(float(p.grad.abs().max()) for p in net_b.parameters())
        for k, p in net_b.named_parameters():
            log.append((s, k, rel_l2(ga[k], p.grad, gmax), rel_l2(gc_[k], p.grad, gmax),
                        abs(float(loss_a) - float(loss_b)) / abs(float(loss_b)),
                        abs(float(loss_c) - float(loss_b)) / abs(float(loss_b)),
                        bool(torch.equal(ga[k], p.grad)), bool(torch.equal(gc_[k], p.grad))))
    assert graphed.stats["captures"] == 1 and graphed.stats["replays"] == 3


def main():
    from boostmvsnerfs_amd import _lib
    from boostmvsnerfs_amd.config import set_cfg
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(args[0]) if args else 50
    if "--det" in sys.argv:
        _lib.set_tuning("BMV_DETERMINISTIC", 1)
    enerf_fx = load_fixture("enerf_tiny")
    set_cfg(tiny_cfg(enerf_fx, "enerf_pretrain"))
    log = []
    for rep in range(reps):
        n0 = len(log)
        one_rep(enerf_fx, log)
        rows = log[n0:]
        wa = max(rows, key=lambda r: r[2])
        we = max(rows, key=lambda r: r[3])
        print(f"rep {rep:3d}  graphed-vs-eager worst {wa[2]:.3e} (step {wa[0]} {wa[1]})   eager-vs-eager worst {we[3]:.3e} "
              f"(step {we[0]} {we[1]})   bit-equal tensors g/e {sum(r[6] for r in rows)}/{sum(r[7] for r in rows)} of {len(rows)}",
              flush=True)
    kinds = {"eager steps 0-2": (0, 1, 2), "capture + first replay (3)": (3,), "replays (4, 5)": (4, 5)}
    print("\nper step kind, over all tensors and repetitions: relative L2 of the gradient, graphed-vs-eager | eager-vs-eager")
    for name, steps in kinds.items():
        for col, label in ((2, "A-B"), (3, "B-C")):
            v = torch.tensor([r[col] for r in log if r[0] in steps], dtype=torch.float64)
            print(f"  {name:28s} {label}: n {v.numel():6d}  mean {float(v.mean()):.3e}  p50 {float(v.median()):.3e}  "
                  f"p99 {float(v.quantile(0.99)):.3e}  p99.9 {float(v.quantile(0.999)):.3e}  max {float(v.max()):.3e}")
    print("\nper tensor (worst 12 by eager-vs-eager max): max A-B on replays | max B-C on any step | mean B-C | std B-C")
    per = {}
    for r in log:
        d = per.setdefault(r[1], {"ab": [], "bc": []})
        if r[0] >= 3:
            d["ab"].append(r[2])
        d["bc"].append(r[3])
    rows = sorted(per.items(), key=lambda kv: -max(kv[1]["bc"]))[:12]
    for k, d in rows:
        bc = torch.tensor(d["bc"], dtype=torch.float64)
        print(f"  {k:44s} {max(d['ab']):.3e} | {float(bc.max()):.3e} | {float(bc.mean()):.3e} | {float(bc.std()):.3e}")
    lo = max(r[4] for r in log), max(r[5] for r in log)
    print(f"\nloss: worst relative difference graphed-vs-eager {lo[0]:.3e}, eager-vs-eager {lo[1]:.3e}")
    # the bar a statistical test could use: mean + 6 sigma of the eager-vs-eager distribution of the WORST tensor per step
    worst = {}
    for i, r in enumerate(log):
        key = (i // (115 * 6), r[0])
        worst[key] = max(worst.get(key, 0.0), r[3])
    w = torch.tensor(list(worst.values()), dtype=torch.float64)
    print(f"worst-tensor-per-step eager-vs-eager: mean {float(w.mean()):.3e} std {float(w.std()):.3e} "
          f"mean + 6 sigma {float(w.mean() + 6 * w.std()):.3e} max {float(w.max()):.3e}")


if __name__ == "__main__":
    main()
