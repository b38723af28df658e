"""The split-bf16 3x3x3 convolution (csrc/conv_split.hip, experiment) against the fp32 engine on the regularisers' first
layers and heads of the 512x640 frame."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import convnet  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for name, cin, cout, dhw in (("level-1 conv0", 16, 8, (8, 256, 320)), ("level-0 conv0", 32, 8, (64, 64, 80)),
                                 ("level-1 heads", 8, 9, (8, 256, 320)), ("level-0 heads", 8, 9, (64, 64, 80))):
        x = torch.randn(1, cin, *dhw, device=dev)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
        b = torch.randn(cout, device=dev)
        p32 = convnet.pack_conv(w, b)
        o32 = torch.empty(1, cout, *dhw, device=dev)
        t32 = timeit(lambda: convnet.conv_fwd(x, *p32, cout, 3, 3, relu=True, out=o32))
        line = f"{name:15s} {cin:2d}->{cout:2d} {dhw}: fp32 engine {t32:6.1f} us"
        for parts in (3, 2):
            ps = convnet.pack_conv_split(w, b, parts=parts)
            os_ = torch.empty_like(o32)
            ts = timeit(lambda: convnet.conv3d_split_fwd(x, *ps, cout, relu=True, out=os_))
            line += f"   {parts}-piece split {ts:6.1f} us (max |d| {float((o32 - os_).abs().max()):.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
