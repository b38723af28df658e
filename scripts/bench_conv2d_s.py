"""FeatureNet's encoder layers at the frame's sizes: the fp32 engine (csrc/conv.hip) against the bf16 x 3 strip walk
(csrc/conv2d_s.hip), per row tiling; HIP-graph timed.
    python scripts/bench_conv2d_s.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from boostmvsnerfs_amd import _lib, convnet  # noqa: E402
from bench_conv_c4 import timed  # noqa: E402

LAYERS = [("conv1.0", 8, 16, 5, 2, 1), ("conv1.1", 16, 16, 3, 1, 2), ("conv2.0", 16, 32, 5, 2, 2), ("conv2.1", 32, 32, 3, 1, 4)]


def main():
    for B, H, W in ((3, 512, 640), (6, 480, 736)):
        for name, cin, cout, ks, stride, div in LAYERS:
            g = torch.Generator().manual_seed(0)
            x = torch.randn(B, cin, H // div, W // div, generator=g).cuda()
            w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).cuda()
            b = torch.randn(cout, generator=g).cuda()
            wp, bp = convnet.pack_conv(w, b, stride=stride)
            t32 = timed(lambda: convnet.conv_fwd(x, wp, bp, cout, 1, ks, stride, relu=True))
            ws, bs = convnet.pack_conv2d_s(w, b)
            line = f"{B} x {H} x {W} {name} ({cin}->{cout} k{ks}s{stride} on {H // div}x{W // div}): fp32 engine {t32:6.1f} us | conv2d_s:"
            for rows in (0, 4, 8):
                _lib.set_tuning("BMV_CONV2D_S_ROWS", rows)
                line += f"  rows {rows}: {timed(lambda: convnet.conv2d_s(x, ws, bs, cout, ks, stride, relu=True)):6.1f}"
            xr = convnet.SplitRecords.from_planar(x)
            line += " | records in:"
            for rows in (4, 8):
                _lib.set_tuning("BMV_CONV2D_S_ROWS", rows)
                line += f"  rows {rows}: {timed(lambda: convnet.conv2d_s(xr, ws, bs, cout, ks, stride, relu=True)):6.1f}"
            _lib.set_tuning("BMV_CONV2D_S_ROWS", 4)
            line += f" | rows 4, records in + out: {timed(lambda: convnet.conv2d_s(xr, ws, bs, cout, ks, stride, relu=True, records=True)):6.1f}"
            line += f", in + both: {timed(lambda: convnet.conv2d_s(xr, ws, bs, cout, ks, stride, relu=True, records='both')):6.1f}"
            line += f", planar in, records out: {timed(lambda: convnet.conv2d_s(x, ws, bs, cout, ks, stride, relu=True, records=True)):6.1f}"
            _lib.set_tuning("BMV_CONV2D_S_ROWS", None)
            print(line, flush=True)


if __name__ == "__main__":
    main()
