"""Where the time of the bf16 x 3 encoder convolutions (csrc/conv2d_s.hip) goes: ablation BUILDS (-DBMV_C2S_ABLATE=n:
results are wrong, timing only) at 3 x 512 x 640, HIP-graph timed; every variant compiled to its own library under /tmp.
    python scripts/ablate_conv2d_s.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "boostmvsnerfs_amd", "csrc")
FLAGS = [(0, "full kernel"), (1, "no matrix instructions"), (2, "no loads"), (4, "no split / LDS writes"), (8, "no stores"),
         (2 | 4 | 8, "matrix + LDS reads only"), (1 | 8, "staging only"), (1 | 2 | 4, "stores only"), (1 | 2 | 4 | 8, "skeleton")]


def child():
    import torch
    from boostmvsnerfs_amd import convnet
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_conv_c4 import timed
    out = []
    for cin, cout, ks, stride, div in ((8, 16, 5, 2, 1), (16, 16, 3, 1, 2), (16, 32, 5, 2, 2)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(3, cin, 512 // div, 640 // div, generator=g).cuda()
        w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).cuda()
        b = torch.randn(cout, generator=g).cuda()
        ws, bs = convnet.pack_conv2d_s(w, b)
        out.append(timed(lambda: convnet.conv2d_s(x, ws, bs, cout, ks, stride, relu=True)))
    print("TIMES %.2f %.2f %.2f" % tuple(out))


def main():
    from boostmvsnerfs_amd import build
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in build.SOURCES if s != "conv2d_s.hip"]
    print(f"{'flags':>5s}  {'build':28s} {'conv1.0':>10s} {'conv1.1':>10s} {'conv2.0':>10s}   (us, 3 x 512 x 640 frame)")
    for fl, what in FLAGS:
        o, lib = f"/tmp/conv2d_s_ab{fl}.o", f"/tmp/libbmv_conv2d_s_ab{fl}.so"
        subprocess.check_call([build._hipcc(), *build.FLAGS, f"-DBMV_C2S_ABLATE={fl}", "-c", os.path.join(CSRC, "conv2d_s.hip"), "-o", o])
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, o])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, BMV_LIB_PATH=lib),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("TIMES ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(1)
        a, b, c = line[0].split()[1:]
        print(f"{fl:5d}  {what:28s} {float(a):10.1f} {float(b):10.1f} {float(c):10.1f}", flush=True)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
