"""The 4 x 4 x 1-block convolution kernel (csrc/conv_c4.hip) against the 16-row engine kernels it would replace, on the
four layer shapes of the headline frame that have <= 9 output channels; HIP-graph timed (50 launches per replay).

    python scripts/bench_conv_c4.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import convnet  # noqa: E402

DEV = "cuda"


def timed(fn, n=50, reps=5):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    shapes = [("L0 conv0 32->8", 3, 1, 32, 8, (64, 64, 80)), ("L1 conv0 16->8", 3, 1, 16, 8, (8, 256, 320)),
              ("L1 heads 8->9", 3, 1, 8, 9, (8, 256, 320)), ("L0 heads 8->9", 3, 1, 8, 9, (64, 64, 80)),
              ("smooth0 32->8 2-D", 2, 3, 32, 8, (512, 640)), ("smooth1 32->16?", 2, 3, 32, 8, (256, 320))]
    for name, nd, B, Cin, Cout, sp in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, Cin, *sp, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, *((3,) * nd), generator=g) / (Cin * 3 ** nd) ** 0.5).to(DEV)
        b = torch.randn(Cout, generator=g).to(DEV)
        wp16, bp16 = convnet.pack_conv(w, b, 1)
        wp4, bp4 = convnet.pack_conv_c4(w, b)
        flops = 2.0 * x.numel() / Cin * Cout * Cin * 3 ** nd
        t16 = timed(lambda: convnet.conv_fwd(x, wp16, bp16, Cout, 3 if nd == 3 else 1, 3, 1, relu=True))
        line = f"{name:22s} {flops / 1e9:5.2f} GF  engine {t16:7.1f} us ({flops / t16 / 1e6:5.1f} TF/s)"
        for v in ((0, 1, 2, 4) if nd == 3 else (0, 2)):
            try:
                t4 = timed(lambda: convnet.conv_c4_fwd(x, wp4, bp4, Cout, relu=True, variant=v))
                line += f"  c4 v{v} {t4:6.1f} us ({flops / t4 / 1e6:5.1f})"
            except RuntimeError as e:
                line += f"  c4 v{v} n/a"
        if nd == 3 and Cin % 4 == 0:      # the input as quad records (what the sweep / conv11 hand over in the frame)
            from boostmvsnerfs_amd import ops
            D_, H_, W_ = sp
            qv = ops.QuadVolume(x.view(B, Cin // 4, 4, D_, H_, W_).permute(0, 1, 3, 4, 5, 2).contiguous())
            t4 = timed(lambda: convnet.conv_c4_fwd(qv, wp4, bp4, Cout, relu=True))
            line += f"  c4 v0, quad-record input {t4:6.1f} us ({flops / t4 / 1e6:5.1f})"
            if Cin % 8 == 0:                # ... and the bf16 x 3 form of the same layer (csrc/conv_c4s.hip)
                ws, bs, pr = convnet.pack_conv_c4s(w, b)
                ts = timed(lambda: convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, relu=True, quad_out=Cout % 4 == 0))
                line += f"  c4s (bf16 x 3{', paired' if pr else ''}) {ts:6.1f} us ({flops / ts / 1e6:5.1f})"
                if Cout == 9:
                    ws, bs, pr = convnet.pack_conv_c4s(w, None)
                    ts = timed(lambda: convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, records=True))
                    t4 = timed(lambda: convnet.conv_c4_fwd(qv, wp4, bp4, Cout, records=True))
                    line += f"  records: c4 {t4:6.1f} / c4s {ts:6.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
