"""Where the time of the 4-row-block convolution kernel (csrc/conv_c4.hip) goes: ablation BUILDS (-DBMV_C4_ABLATE=n:
results are wrong, timing only) of the frame's four 3-D layers, HIP-graph timed.  Every variant is compiled to its own
library under /tmp (conv_c4.hip with the define + the tree's other objects) and timed in a child process.

    python scripts/ablate_conv_c4.py            # on the MI355X box (hipcc is in the image)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "boostmvsnerfs_amd", "csrc")
FLAGS = [(0, "full kernel"), (1, "no matrix instructions"), (2, "no tile loads"), (4, "no LDS reads per tap"), (8, "no weight copy"),
         (16, "no stores"), (2 | 8 | 16, "matrix + LDS only"), (2 | 4 | 8 | 16, "matrix instructions only"),
         (1 | 4 | 16, "loads + staging only"), (1 | 2 | 4 | 8 | 16, "skeleton")]
SHAPES = [("L0 conv0 32->8", 1, 32, 8, (64, 64, 80)), ("L1 conv0 16->8", 1, 16, 8, (8, 256, 320)),
          ("L1 heads 8->9", 1, 8, 9, (8, 256, 320)), ("L0 heads 8->9", 1, 8, 9, (64, 64, 80))]


def child():
    import torch
    from boostmvsnerfs_amd import convnet
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_conv_c4 import timed
    out = []
    for name, B, Cin, Cout, sp in SHAPES:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, Cin, *sp, generator=g).to("cuda")
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).to("cuda")
        b = torch.randn(Cout, generator=g).to("cuda")
        wp4, bp4 = convnet.pack_conv_c4(w, b)
        out.append(timed(lambda: convnet.conv_c4_fwd(x, wp4, bp4, Cout, relu=True, variant=0)))
    print("TIMES " + " ".join(f"{t:.2f}" for t in out))


def main():
    from boostmvsnerfs_amd import build
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in build.SOURCES if s != "conv_c4.hip"]
    rows = []
    for fl, what in FLAGS:
        o, lib = f"/tmp/conv_c4_ab{fl}.o", f"/tmp/libbmv_c4_ab{fl}.so"
        subprocess.check_call([build._hipcc(), *build.FLAGS, f"-DBMV_C4_ABLATE={fl}", "-c", os.path.join(CSRC, "conv_c4.hip"), "-o", o])
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, o])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, BMV_LIB_PATH=lib),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("TIMES ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(1)
        rows.append((fl, what, [float(t) for t in line[0].split()[1:]]))
    print(f"{'flags':>5s}  {'build':28s} " + " ".join(f"{n:>16s}" for n, *_ in SHAPES) + "   (us; 4x4x1 MFMA time at 122 TFLOP/s: 37.1 37.1 20.9 10.4)")
    for fl, what, ts in rows:
        print(f"{fl:5d}  {what:28s} " + " ".join(f"{t:16.1f}" for t in ts), flush=True)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
