"""FeatureNet's fused top-down + smooth0 launch at the frame's size: the fp32 kernel (csrc/conv.hip fpn_smooth_kernel)
against the bf16 x 3 kernel with lat0 folded into the weights (csrc/fpn_s.hip), per row tiling; HIP-graph timed.
    python scripts/bench_fpn_smooth.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from boostmvsnerfs_amd import _lib, convnet  # noqa: E402
from bench_conv_c4 import timed  # noqa: E402


def main():
    for B, H, W in ((3, 512, 640), (6, 480, 736), (3, 256, 320)):
        g = torch.Generator().manual_seed(0)
        fine = torch.randn(B, 8, H, W, generator=g).cuda()
        coarse = torch.randn(B, 32, H // 2, W // 2, generator=g).cuda()
        wl, bl = torch.randn(32, 8, 1, 1, generator=g).cuda(), torch.randn(32, generator=g).cuda()
        ws, bs = (torch.randn(8, 32, 3, 3, generator=g) / 6).cuda(), torch.randn(8, generator=g).cuda()
        rgb = torch.rand(B, 3, H, W, generator=g).cuda()
        eo = convnet.LookupRecords.EVEN_ODD
        wp, bp = convnet.pack_conv(ws[list(eo)], bs[list(eo)])
        t32 = timed(lambda: convnet.fpn_smooth(fine, coarse, wl, bl, wp, bp, 8, rgb=rgb))
        wsp, bt = convnet.pack_fpn_smooth_s(ws, bs, wl, bl, order=eo)
        line = f"{B} x {H} x {W}: fp32 kernel {t32:6.1f} us | bf16 x 3, folded:"
        for rows in (0, 6, 8, 9, 10, 12):
            _lib.set_tuning("BMV_FPN_S_ROWS", rows)
            line += f"  rows {rows}: {timed(lambda: convnet.fpn_smooth_s(fine, coarse, wsp, bt, rgb=rgb)):6.1f}"
        _lib.set_tuning("BMV_FPN_S_ROWS", None)
        print(line, flush=True)
        # FeatureNet's first block: the fp32 fused kernel against conv0_s (second layer on the bf16 matrix cores)
        x = torch.randn(B, 3, H, W, generator=g).cuda()
        w0, b0 = (torch.randn(8, 3, 3, 3, generator=g) / 3).cuda(), torch.randn(8, generator=g).cuda()
        w1, b1 = (torch.randn(8, 8, 3, 3, generator=g) / 6).cuda(), torch.randn(8, generator=g).cuda()
        wp1, bp1 = convnet.pack_conv(w1, b1)
        t32 = timed(lambda: convnet.conv0_fused(x, w0, b0, wp1, bp1, 8))
        pk = convnet.pack_conv0_s(w0, b0, w1, b1)
        line = f"{B} x {H} x {W}: conv0 block, fp32 kernel {t32:6.1f} us | conv0_s:"
        for rows in (0, 9, 10, 12):
            _lib.set_tuning("BMV_CONV0_S_ROWS", rows)
            line += f"  rows {rows}: {timed(lambda: convnet.conv0_s(x, *pk)):6.1f}"
        _lib.set_tuning("BMV_CONV0_S_ROWS", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
