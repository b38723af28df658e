"""Where does run-to-run nondeterminism enter a training step?  Runs forward + loss + backward of the tiny ENeRF fixture
TWICE from the same state with every function of boostmvsnerfs_amd.ops wrapped: the tensors each call reads and returns
are fingerprinted (exact: the float bits summed as int64, order-independent and reproducible), the two traces are
compared call by call, and the first calls whose INPUTS agree while their OUTPUTS differ are printed -- those are the
nondeterministic kernels; a call whose inputs already differ inherited it from a torch op in between (printed too).

    python tests/tools/det_trace.py [--det] [--boost]
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


def fp(t):
    if not torch.is_tensor(t) or not t.is_cuda:
        return None
    x = t.detach().contiguous()
    if x.dtype == torch.float32:
        x = x.view(torch.int32)
    elif x.dtype not in (torch.int32, torch.int64):
        return None
    return int(x.to(torch.int64).sum()), tuple(t.shape)


def flat(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            yield from flat(y)
    elif hasattr(x, "t") and torch.is_tensor(getattr(x, "t")):
        yield x.t


def main():
    from boostmvsnerfs_amd import _lib, ops
    from boostmvsnerfs_amd.config import set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.train import NetworkWrapper
    if "--det" in sys.argv:
        _lib.set_tuning("BMV_DETERMINISTIC", 1)
    enerf_fx = load_fixture("enerf_tiny")
    set_cfg(tiny_cfg(enerf_fx, "enerf_pretrain"))
    net0 = Network()
    net0.load_state_dict(enerf_fx.group("sd"), strict=True)
    net0 = net0.to(DEV).train()
    base = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in enerf_fx.batch().items()}
    g = torch.Generator().manual_seed(0)
    for i in range(2):
        base[f"rgb_{i}"] = torch.rand(1, base[f"rays_{i}"].shape[1], 3, generator=g).to(DEV)

    trace = []
    originals = {}
    for name in dir(ops):
        fn = getattr(ops, name)
        if callable(fn) and not name.startswith("_") and getattr(fn, "__module__", "") == ops.__name__ and not isinstance(fn, type):
            originals[name] = fn

            def wrapped(*a, _fn=fn, _name=name, **kw):
                ins = [fp(t) for t in flat(list(a) + list(kw.values()))]
                out = _fn(*a, **kw)
                trace.append((_name, ins, [fp(t) for t in flat(out)]))
                return out
            setattr(ops, name, wrapped)

    runs = []
    grads = []
    for rep in range(2):
        net = copy.deepcopy(net0).train()
        trace.clear()
        _, loss, _, _ = NetworkWrapper(net)(dict(base))
        loss.mean().backward()
        torch.cuda.synchronize()
        runs.append(list(trace))
        grads.append({k: p.grad.clone() for k, p in net.named_parameters()})
    a, b = runs
    print(f"{len(a)} / {len(b)} traced calls; loss bits equal")
    shown = 0
    for i, (ca, cb) in enumerate(zip(a, b)):
        assert ca[0] == cb[0], (i, ca[0], cb[0])
        same_in, same_out = ca[1] == cb[1], ca[2] == cb[2]
        if not same_out and shown < 25:
            print(f"  call {i:4d} {ca[0]:32s} inputs {'EQUAL' if same_in else 'differ'}  outputs differ "
                  f"{[s[1] for s, t in zip(ca[2], cb[2]) if s != t]}")
            shown += 1
    bad = [k for k in grads[0] if not torch.equal(grads[0][k], grads[1][k])]
    print(f"parameter gradients that differ between the two runs: {len(bad)} of {len(grads[0])}")
    for k in bad[:20]:
        print("   ", k)


if __name__ == "__main__":
    main()
