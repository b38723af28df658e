"""Where the time of the bf16 x 3 first-layer / heads kernel (csrc/conv_c4s.hip) goes: ablation BUILDS (-DBMV_C4S_ABLATE=n:
results are wrong, timing only) of the frame's four 3-D layers, HIP-graph timed.  Every variant is compiled to its own
library under /tmp (conv_c4s.hip with the define + the tree's other objects) and timed in a child process.

    python scripts/ablate_conv_c4s.py [extra -D flags]           # on the MI355X box (hipcc is in the image)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "boostmvsnerfs_amd", "csrc")
FLAGS = [(0, "full kernel"), (1, "no matrix instructions"), (2, "no plane loads"), (4, "no split / LDS writes"), (8, "no A loads"),
         (16, "no stores"), (32, "no per-plane barrier"), (2 | 4, "no staging at all"), (2 | 4 | 8 | 16, "matrix + LDS reads only"),
         (1 | 8 | 16, "staging only"), (1 | 2 | 4 | 8 | 16, "skeleton")]
SHAPES = [("L0 conv0 32->8", 1, 32, 8, (64, 64, 80)), ("L1 conv0 16->8", 1, 16, 8, (8, 256, 320)),
          ("L1 heads 8->9", 1, 8, 9, (8, 256, 320)), ("L0 heads 8->9", 1, 8, 9, (64, 64, 80))]


def child():
    import torch
    from boostmvsnerfs_amd import convnet, ops
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_conv_c4 import timed
    out = []
    for name, B, Cin, Cout, sp in SHAPES:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, Cin, *sp, generator=g).to("cuda")
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).to("cuda")
        b = torch.randn(Cout, generator=g).to("cuda")
        D_, H_, W_ = sp
        qv = ops.QuadVolume(x.view(B, Cin // 4, 4, D_, H_, W_).permute(0, 1, 3, 4, 5, 2).contiguous())
        ws, bs, pr = convnet.pack_conv_c4s(w, b)
        out.append(timed(lambda: convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, relu=True, quad_out=Cout % 4 == 0, records=Cout == 9)))
    print("TIMES " + " ".join(f"{t:.2f}" for t in out))


def main():
    from boostmvsnerfs_amd import build
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    only = [int(a) for a in sys.argv[1:] if a.isdigit()]
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in build.SOURCES if s != "conv_c4s.hip"]
    rows = []
    for fl, what in FLAGS:
        if only and fl not in only:
            continue
        o, lib = f"/tmp/conv_c4s_ab{fl}.o", f"/tmp/libbmv_c4s_ab{fl}.so"
        subprocess.check_call([build._hipcc(), *build.FLAGS, f"-DBMV_C4S_ABLATE={fl}", *extra, "-c", os.path.join(CSRC, "conv_c4s.hip"), "-o", o])
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, o])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, BMV_LIB_PATH=lib),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("TIMES ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(1)
        rows.append((fl, what, [float(t) for t in line[0].split()[1:]]))
    print(f"{'flags':>5s}  {'build':28s} " + " ".join(f"{n:>16s}" for n, *_ in SHAPES) + "   (us; bf16 x 3 MFMA time at 2.4 GHz: 14.4 14.4 14.4 7.2)")
    for fl, what, ts in rows:
        print(f"{fl:5d}  {what:28s} " + " ".join(f"{t:16.1f}" for t in ts), flush=True)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
