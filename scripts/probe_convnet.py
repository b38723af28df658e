"""Per-layer timing of the convolution engine against torch/MIOpen at the layer shapes of BASELINE
configs[1] (ENeRF 512x640, 3 source views, planes [64, 8]).  Prints one line per layer:
    name  ours_us  miopen_us  GFLOP  ours TFLOP/s
Run on the MI355X box:  python scripts/probe_convnet.py [--find 1]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import convnet  # noqa: E402

LAYERS = [
    # name, dims, B, Cin, Cout, k, stride, (D,) H, W
    ("feat.conv0.0", 2, 3, 3, 8, 3, 1, 512, 640), ("feat.conv0.1", 2, 3, 8, 8, 3, 1, 512, 640),
    ("feat.conv1.0", 2, 3, 8, 16, 5, 2, 512, 640), ("feat.conv1.1", 2, 3, 16, 16, 3, 1, 256, 320),
    ("feat.conv2.0", 2, 3, 16, 32, 5, 2, 256, 320), ("feat.conv2.1", 2, 3, 32, 32, 3, 1, 128, 160),
    ("feat.toplayer", 2, 3, 32, 32, 1, 1, 128, 160), ("feat.smooth1", 2, 3, 32, 16, 3, 1, 256, 320),
    ("feat.smooth0", 2, 3, 32, 8, 3, 1, 512, 640),
    ("reg0.conv0", 3, 1, 32, 8, 3, 1, 64, 64, 80), ("reg0.conv1", 3, 1, 8, 16, 3, 2, 64, 64, 80),
    ("reg0.conv2", 3, 1, 16, 16, 3, 1, 32, 32, 40), ("reg0.conv3", 3, 1, 16, 32, 3, 2, 32, 32, 40),
    ("reg0.conv4", 3, 1, 32, 32, 3, 1, 16, 16, 20), ("reg0.feat+depth", 3, 1, 8, 9, 3, 1, 64, 64, 80),
    ("reg1.conv0", 3, 1, 16, 8, 3, 1, 8, 256, 320), ("reg1.conv1", 3, 1, 8, 16, 3, 2, 8, 256, 320),
    ("reg1.conv2", 3, 1, 16, 16, 3, 1, 4, 128, 160), ("reg1.conv3", 3, 1, 16, 32, 3, 2, 4, 128, 160),
    ("reg1.conv4", 3, 1, 32, 32, 3, 1, 2, 64, 80), ("reg1.conv5", 3, 1, 32, 64, 3, 2, 2, 64, 80),
    ("reg1.conv6", 3, 1, 64, 64, 3, 1, 1, 32, 40), ("reg1.feat+depth", 3, 1, 8, 9, 3, 1, 8, 256, 320),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--find", type=int, default=1)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = bool(args.find)
    tot_o = tot_m = 0.0
    for L in LAYERS:
        name, nd, B, Cin, Cout, k, s = L[:7]
        if args.only and args.only not in name:
            continue
        sp = L[7:]
        x = torch.randn(B, Cin, *sp, device="cuda")
        w = torch.randn(Cout, Cin, *([k] * nd), device="cuda") / (Cin * k ** nd) ** 0.5
        b = torch.randn(Cout, device="cuda")
        wp, bp = convnet.pack_conv(w, b, s)
        kd = k if nd == 3 else 1
        conv = F.conv3d if nd == 3 else F.conv2d
        with torch.no_grad():
            want = F.relu(conv(x, w, b, s, k // 2))
            got = convnet.conv_fwd(x, wp, bp, Cout, kd, k, s, relu=True)
            err = float((got - want).abs().max()) / float(want.abs().max())
            t_o = timeit(lambda: convnet.conv_fwd(x, wp, bp, Cout, kd, k, s, relu=True, out=got))
            t_m = timeit(lambda: F.relu(conv(x, w, b, s, k // 2), inplace=True))
        gf = 2.0 * want.numel() * Cin * k ** nd / 1e9
        tot_o, tot_m = tot_o + t_o, tot_m + t_m
        print(f"{name:18s} ours {t_o:8.1f} us   miopen+relu {t_m:8.1f} us   {gf:6.2f} GFLOP  {gf / t_o * 1e3:6.1f} TFLOP/s"
              f"   rel err {err:.1e}", flush=True)
    print(f"total ours {tot_o:.0f} us, miopen {tot_m:.0f} us")


if __name__ == "__main__":
    main()
