"""The bf16 x 3 first-layer / heads kernel (csrc/conv_c4s.hip) over its tilings (rows per wave RW x planes per workgroup TZ)
on the frame's four layers, against the fp32 4-row-block kernel (csrc/conv_c4.hip) on the same quad-record input; HIP-graph
timed (50 launches per replay).      python scripts/bench_conv_c4s.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from boostmvsnerfs_amd import _lib, convnet, ops  # noqa: E402
from bench_conv_c4 import timed  # noqa: E402


def main():
    shapes = [("L0 conv0 32->8", 1, 32, 8, (64, 64, 80)), ("L1 conv0 16->8", 1, 16, 8, (8, 256, 320)),
              ("L1 heads 8->9", 1, 8, 9, (8, 256, 320)), ("L0 heads 8->9", 1, 8, 9, (64, 64, 80)),
              ("cfg3 L1 conv0 16->8", 1, 16, 8, (8, 240, 368)), ("cfg1 L1 conv0 16->8", 1, 16, 8, (8, 128, 160))]
    for name, B, Cin, Cout, sp in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, Cin, *sp, generator=g).to("cuda")
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).to("cuda")
        b = torch.randn(Cout, generator=g).to("cuda")
        D_, H_, W_ = sp
        qv = ops.QuadVolume(x.view(B, Cin // 4, 4, D_, H_, W_).permute(0, 1, 3, 4, 5, 2).contiguous())
        wp4, bp4 = convnet.pack_conv_c4(w, b)
        ws, bs, pr = convnet.pack_conv_c4s(w, b)
        rec = Cout == 9
        t4 = timed(lambda: convnet.conv_c4_fwd(qv, wp4, bp4, Cout, relu=True, records=rec, quad_out=not rec))
        line = f"{name:22s} c4 (fp32) {t4:6.1f} us | c4s auto "
        _lib.set_tuning("BMV_CONV_C4S_RW", None), _lib.set_tuning("BMV_CONV_C4S_TZ", None)
        line += f"{timed(lambda: convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, relu=True, records=rec, quad_out=not rec)):6.1f} |"
        for rw in (4, 2):
            for tz in (4, 2):
                if not pr and rw == 4 and tz == 4:
                    continue
                _lib.set_tuning("BMV_CONV_C4S_RW", rw), _lib.set_tuning("BMV_CONV_C4S_TZ", tz)
                ts = timed(lambda: convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, relu=True, records=rec, quad_out=not rec))
                line += f"  RW{rw} TZ{tz} {ts:6.1f}"
        _lib.set_tuning("BMV_CONV_C4S_RW", None), _lib.set_tuning("BMV_CONV_C4S_TZ", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
