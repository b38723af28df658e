"""Where the time of the 4-row-block convolution kernel (csrc/conv_c4.hip) goes: its ablation flags (bmv_tuning
BMV_CONV_C4_FLAGS: results are wrong, timing only) on the frame's four 3-D layers, HIP-graph timed.

    python scripts/ablate_conv_c4.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import _lib, convnet  # noqa: E402
from bench_conv_c4 import timed  # noqa: E402

DEV = "cuda"
FLAGS = [(0, "full kernel"), (1, "no matrix instructions"), (2, "no tile loads"), (4, "no LDS reads per tap"), (8, "no weight copy"),
         (16, "no stores"), (2 | 8 | 16, "matrix + LDS only"), (2 | 4 | 8 | 16, "matrix instructions only"),
         (1 | 4 | 16, "loads + staging only"), (1 | 2 | 4 | 8 | 16, "skeleton")]


def main():
    shapes = [("L0 conv0 32->8", 1, 32, 8, (64, 64, 80)), ("L1 conv0 16->8", 1, 16, 8, (8, 256, 320)),
              ("L1 heads 8->9", 1, 8, 9, (8, 256, 320)), ("L0 heads 8->9", 1, 8, 9, (64, 64, 80))]
    for name, B, Cin, Cout, sp in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, Cin, *sp, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).to(DEV)
        b = torch.randn(Cout, generator=g).to(DEV)
        wp4, bp4 = convnet.pack_conv_c4(w, b)
        flops = 2.0 * x.numel() * Cout * 27
        print(f"{name}  {flops / 1e9:.2f} GF  ({flops / 122e6:.1f} us of 4x4x1 MFMAs at the measured 122 TFLOP/s)")
        for fl, what in FLAGS:
            _lib.set_tuning("BMV_CONV_C4_FLAGS", fl)
            t = timed(lambda: convnet.conv_c4_fwd(x, wp4, bp4, Cout, relu=True, variant=0))
            print(f"    flags {fl:2d}  {what:28s} {t:7.1f} us", flush=True)
        _lib.set_tuning("BMV_CONV_C4_FLAGS", 0)


if __name__ == "__main__":
    main()
