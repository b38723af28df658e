"""Would the regularisers' 16 -> 16 interior layer (conv2, half resolution) run faster on csrc/conv_c4s.hip's unpaired bf16 x 3
kernel (quad records in / out) than on the fp32 engine?  Stand-alone, HIP-graph timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from boostmvsnerfs_amd import convnet, ops  # noqa: E402
from bench_conv_c4 import timed  # noqa: E402

for name, (B, C, D, H, W) in (("level 1 conv2", (1, 16, 4, 128, 160)), ("level 0 conv2", (1, 16, 32, 32, 40)),
                              ("config 3 level 1 conv2", (1, 16, 4, 120, 184))):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, D, H, W, generator=g).cuda()
    w = (torch.randn(16, C, 3, 3, 3, generator=g) / (C * 27) ** 0.5).cuda()
    b = torch.randn(16, generator=g).cuda()
    wp, bp = convnet.pack_conv(w, b)
    t32 = timed(lambda: convnet.conv_fwd(x, wp, bp, 16, 3, 3, 1, relu=True))
    xq = ops.QuadVolume(x.view(B, C // 4, 4, D, H, W).permute(0, 1, 3, 4, 5, 2).contiguous())
    ws, bs, pr = convnet.pack_conv_c4s(w, b, False)
    ts = timed(lambda: convnet.conv_c4s_fwd(xq, ws, bs, pr, 16, relu=True, quad_out=True))
    got = convnet.conv_c4s_fwd(xq, ws, bs, pr, 16, relu=True, quad_out=True).to_planar()
    want = convnet.conv_fwd(x, wp, bp, 16, 3, 3, 1, relu=True)
    print(f"{name} {C}->16 on {D}x{H}x{W}: fp32 engine {t32:6.1f} us, conv_c4s unpaired {ts:6.1f} us, max |d| {float((got - want).abs().max()):.2e}")
