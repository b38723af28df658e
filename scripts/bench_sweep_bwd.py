"""sweep backward (d variance -> d source features, d hypotheses) alone, both cascade levels of the 512x640 frame."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    torch.manual_seed(0)
    H, W = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (512, 640)))
    from boostmvsnerfs_amd.synthetic import make_batch
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    cfg = set_cfg(make_cfg("enerf_eval"))
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(H, W, n_views=3, seed=0).items()}
    for lvl, (C, D, vs, fs) in enumerate(((32, 64, 0.125, 0.25), (16, 8, 0.5, 0.5))):
        h, w, Hs, Ws = int(H * vs), int(W * vs), int(H * fs), int(W * fs)
        feats = torch.randn(1, 3, C, Hs, Ws, device=dev)
        proj = ops.proj_mats(batch["src_exts"], batch["src_ixts"], batch["tar_ext"], batch["tar_ixt"], fs, vs)
        nf = batch["near_far"]
        if lvl == 0:
            dv, _ = ops.depth_values_uniform(nf, D, h, w, True)
        else:
            h0, w0 = h // 4, w // 4
            depth = torch.full((1, h0, w0), float(nf[0, 0] + nf[0, 1]) / 2, device=dev) * (1 + 0.05 * torch.randn(1, h0, w0, device=dev))
            std = torch.full((1, h0, w0), float(nf[0, 1] - nf[0, 0]) / 40, device=dev)
            dv, _ = ops.depth_values_cascade(depth, std, nf, h, w, D)
        g = torch.randn(1, C, D, h, w, device=dev)
        for algo in ("planar", "cl"):
            for want in (False, True):
                for _ in range(3):
                    ops.sweep_variance_bwd(feats, proj, dv, g, want, algo=algo)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 10
                for _ in range(n):
                    ops.sweep_variance_bwd(feats, proj, dv, g, want, algo=algo)[0].contiguous()
                torch.cuda.synchronize()
                print(f"level {lvl} C={C} D={D} vol {h}x{w} src {Hs}x{Ws} [{algo:6s}] depth-grad={want}: "
                      f"{(time.perf_counter() - t0) / n * 1e6:.0f} us (incl. layout copies)", flush=True)


if __name__ == "__main__":
    main()
