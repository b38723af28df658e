"""Renderer_ours under autograd: the HIP training path (autograd.MvsMLP, csrc/mvs_mlp_train.hip) against nn.Linear + torch
autograd (rocBLAS GEMMs + elementwise kernels) -- forward + backward time per call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd.networks.mvsnerf.network import RendererMLP  # noqa: E402


def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 128
    dev = "cuda"
    torch.manual_seed(0)
    mlp = RendererMLP().to(dev)
    x = torch.randn(npts, 86, device=dev)
    gy = torch.randn(npts, 4, device=dev)
    flops = 2 * npts * sum(p.numel() for n, p in mlp.named_parameters() if n.endswith("weight"))
    for name, fn in (("hip  ", mlp.forward), ("torch", mlp.forward_torch)):
        def step():
            mlp.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_(True)
            fn(xx).backward(gy)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name} forward+backward {npts} points: {dt * 1e3:.3f} ms   ({3 * flops / dt / 1e12:.1f} TFLOP/s of 3x forward FLOPs)"
              f"   peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
