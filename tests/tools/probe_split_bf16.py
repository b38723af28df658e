"""What would a split-precision MFMA path cost in accuracy?  CPU emulation on the oracle (the checker): every convolution
of the network evaluated as three (bf16x3: hi*hi + hi*lo + lo*hi) or six (three-way split) products of bf16 operands
with fp32 accumulation -- exactly what bf16 MFMAs with fp32 accumulators compute, up to summation order -- against the
plain fp32 oracle, on a whole frame with the golden fixtures' sharpened-logit weights.
    python tests/tools/probe_split_bf16.py [H W]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def split(x, parts):
    out, r = [], x
    for _ in range(parts):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def make(conv, parts):
    def f(x, w, b=None, *a, **k):
        xs, ws = split(x, parts), split(w, parts)
        y = None
        for i in range(parts):
            for j in range(parts - i):          # drop the products below 2^-(8 parts) relative
                t = conv(xs[i], ws[j], None, *a, **k)
                y = t if y is None else y + t
        return y if b is None else y + b.view(1, -1, *([1] * (y.dim() - 2)))
    return f


def main():
    H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (128, 160)
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from test_gpu_fullsize import _perturb
    from oracle import enerf as O
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [32, 8]
    set_cfg(cfg)
    torch.manual_seed(0)
    net = _perturb(Network().eval())
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    batch = make_batch(H, W, n_views=3, seed=0)
    with torch.no_grad():
        want = O.enerf_forward(sd, clone_batch(batch), cfg)
    c2, c3, ct3 = F.conv2d, F.conv3d, F.conv_transpose3d
    for parts, name in ((2, "bf16x3 (two-way split, 3 products)"), (3, "bf16x6 (three-way split, 6 products)")):
        F.conv2d, F.conv3d, F.conv_transpose3d = make(c2, parts), make(c3, parts), make(ct3, parts)
        try:
            with torch.no_grad():
                got = O.enerf_forward(sd, clone_batch(batch), cfg)
        finally:
            F.conv2d, F.conv3d, F.conv_transpose3d = c2, c3, ct3
        print(name)
        for k in sorted(want):
            if not torch.is_tensor(want[k]):
                continue
            rms = float(want[k].pow(2).mean().sqrt())
            err = (got[k] - want[k]).abs()
            rel = err / (want[k].abs() + rms)
            tol = 1e-3 * want[k].abs() + 1e-3 * rms
            print(f"   {k:18s} max rel {float(rel.max()):.2e}   mean rel {float(rel.mean()):.2e}   outside the 1e-3 bar: "
                  f"{float((err > tol).float().mean()):.2e}")


if __name__ == "__main__":
    main()
