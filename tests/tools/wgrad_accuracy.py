"""Accuracy of the weight-gradient kernels by themselves (csrc/conv_wgrad.hip) on the shapes of config 5's regularisers:
the SAME fp32 inputs go to (i) bmv_conv_wgrad, (ii) torch's fp32 CPU convolution backward (what the oracle's autograd
runs) and (iii) a float64 evaluation; printed: relative L2 of (i) and (ii) against (iii).

VERDICT r4 read `profiles/r4/grad_fp64_arbitration.txt` (whole-network gradients: HIP 2-4 x farther from float64 than the
fp32 oracle on three cost_reg weights) as a defect of the split-K accumulation; this probe isolates the kernel from what
is upstream of it (the dY it is handed).

    python tests/tools/wgrad_accuracy.py > profiles/r5/wgrad_accuracy.txt
"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from boostmvsnerfs_amd import ops  # noqa: E402


def one(name, B, Cin, Cout, D, H, W, stride, gen):
    dev = torch.device("cuda")
    x = torch.randn(B, Cin, D, H, W, generator=gen).relu_()          # post-ReLU activations: a positive mean
    Do, Ho, Wo = [(n + 2 - 3) // stride + 1 for n in (D, H, W)]
    # dY with a smooth part + noise: neighbouring voxels correlate as a real upstream gradient does
    gy = torch.randn(B, Cout, Do, Ho, Wo, generator=gen) * 1e-3 + 2e-4
    big = F.pad(x, (1, 1 + (stride == 2), 1, 1 + (stride == 2), 1, 1 + (stride == 2)))
    G = ops.conv_wgrad(big.to(dev), gy.to(dev), stride, 3, 3).cpu()
    w = torch.zeros(Cout, Cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x, w, stride=stride, padding=1).backward(gy)
    g32 = w.grad.clone()
    w64 = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), w64, stride=stride, padding=1).backward(gy.double())
    g64 = w64.grad

    def rel(a):
        return float((a.double() - g64).norm() / g64.norm())

    print(f"{name:34s} voxels {B * Do * Ho * Wo:8d}  hip {rel(G):.3e}  torch-cpu-fp32 {rel(g32):.3e}  ratio {rel(G) / max(rel(g32), 1e-30):.2f}")


def main():
    gen = torch.Generator().manual_seed(5)
    # (B = K volumes of config 5 handled one at a time) level 0: 8 -> 64 channels at 32 x 120 x 184 and below;
    # level 1: 8 planes at 240 x 368
    one("cost_reg_0.conv0 8->8", 1, 8, 8, 32, 60, 92, 1, gen)
    one("cost_reg_0.conv1 8->16 s2", 1, 8, 16, 32, 60, 92, 2, gen)
    one("cost_reg_0.conv2 16->16", 1, 16, 16, 16, 30, 46, 1, gen)
    one("cost_reg_0.conv3 16->32 s2", 1, 16, 32, 16, 30, 46, 2, gen)
    one("cost_reg_0.conv4 32->32", 1, 32, 32, 8, 15, 23, 1, gen)
    one("cost_reg_0.conv6 64->64", 1, 64, 64, 4, 8, 12, 1, gen)
    one("cost_reg_1.conv0 8->8", 1, 8, 8, 8, 120, 184, 1, gen)
    one("cost_reg_1.conv2 16->16", 1, 16, 16, 4, 60, 92, 1, gen)
    one("cost_reg_1.conv4 32->32", 1, 32, 32, 2, 30, 46, 1, gen)


if __name__ == "__main__":
    main()
