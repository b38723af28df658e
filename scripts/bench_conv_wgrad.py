"""Time bmv_conv_wgrad on the cost regularisers' layer shapes of the 512x640 fine-tune workload
(level 1: 16 -> 8 volume of 8 x 256 x 320, MinCostRegNet; level 0: 32 -> 8 volume of 64 x 64 x 80, CostRegNet)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import ops  # noqa: E402


def layers(cin, D, H, W, deep):
    L = [("conv0", cin, 8, 1, (D, H, W)), ("conv1", 8, 16, 2, (D, H, W)), ("conv2", 16, 16, 1, (D // 2, H // 2, W // 2)),
         ("conv3", 16, 32, 2, (D // 2, H // 2, W // 2)), ("conv4", 32, 32, 1, (D // 4, H // 4, W // 4))]
    if deep:
        L += [("conv5", 32, 64, 2, (D // 4, H // 4, W // 4)), ("conv6", 64, 64, 1, (D // 8, H // 8, W // 8)),
              ("conv7T", 64, 32, -2, (D // 8, H // 8, W // 8))]
    L += [("conv9T", 32, 16, -2, (D // 4, H // 4, W // 4)), ("conv11T", 16, 8, -2, (D // 2, H // 2, W // 2)),
          ("depth", 8, 1, 1, (D, H, W)), ("feat", 8, 8, 1, (D, H, W))]
    return L


def main():
    dev = "cuda"
    torch.manual_seed(0)
    total = 0.0
    for name, (cin, D, H, W, deep) in (("level1", (16, 8, 256, 320, False)), ("level0", (32, 64, 64, 80, True))):
        for lname, ci, co, stride, (d, h, w) in layers(cin, D, H, W, deep):
            if stride > 0:      # convolution: big = pad(x), small = dY
                x = torch.randn(ci, d, h, w, device=dev)
                od, oh, ow = [(v + 2 - 3) // stride + 1 for v in (d, h, w)]
                small = torch.randn(co, od, oh, ow, device=dev)
                big = F.pad(x, (1, 2 if stride == 2 else 1, 1, 1, 1, 1))
                s = stride
            else:               # transposed: big = pad(dY), small = x
                small = torch.randn(ci, d, h, w, device=dev)
                big = F.pad(torch.randn(co, 2 * d, 2 * h, 2 * w, device=dev), (1, 1, 1, 1, 1, 1))
                s = 2
            for _ in range(2):
                ops.conv_wgrad(big[None], small[None], s, 3, 3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv_wgrad(big[None], small[None], s, 3, 3)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            total += ms
            print(f"{name} {lname:8s} {ci:3d}->{co:3d} s{stride:2d} small {tuple(small.shape)}  {ms * 1000:8.1f} us", flush=True)
    print(f"total {total:.3f} ms")


if __name__ == "__main__":
    main()
