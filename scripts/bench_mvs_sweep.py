"""MVSNeRF's padded plane sweep alone (a19 + a20) at the 224x352 / 128-plane shapes of config 4."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    torch.manual_seed(0)
    H, W, D, pad = 224, 352, 128, 24
    h, w = H // 4, W // 4
    from boostmvsnerfs_amd.synthetic import make_batch
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(H, W, n_views=3, seed=0).items()}
    P = ops.mvs_proj_mats(b["src_exts"], b["src_ixts"])
    small = ops.resize_bilinear(b["src_inps"], h, w)
    feats = torch.randn(1, 3, h, w, 32, device=dev).permute(0, 1, 4, 2, 3)          # channel-last view
    nf = b["near_far"]
    dv = torch.linspace(float(nf[0, 0]), float(nf[0, 1]), D, device=dev)[None]
    for _ in range(3):
        vol = ops.mvs_sweep(small, feats, P, dv, pad)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            ops.mvs_sweep(small, feats, P, dv, pad)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    nbytes = vol.numel() * 4 + feats.numel() * 4 + small.numel() * 4
    print(f"mvs_sweep {tuple(vol.shape)}: {dt * 1e6:.1f} us   {nbytes / dt / 1e9:.0f} GB/s   frac {nbytes / dt / 8e12:.3f}", flush=True)


if __name__ == "__main__":
    main()
