"""Where the time of the bf16 x 3 top-down + smooth0 kernel (csrc/fpn_s.hip) goes: ablation BUILDS (-DBMV_FPN_S_ABLATE=n:
results are wrong, timing only) at 3 x 512 x 640, HIP-graph timed; every variant compiled to its own library under /tmp.
    python scripts/ablate_fpn_s.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "boostmvsnerfs_amd", "csrc")
FLAGS = [(0, "full kernel"), (1, "no matrix instructions"), (2, "no tap / c0 loads"), (4, "no split / LDS writes"), (8, "no stores"),
         (2 | 4 | 8, "matrix + LDS reads only"), (1 | 8, "staging only"), (1 | 2 | 4, "stores only"), (1 | 2 | 4 | 8, "skeleton")]


def child():
    import torch
    from boostmvsnerfs_amd import convnet
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_conv_c4 import timed
    g = torch.Generator().manual_seed(0)
    B, H, W = 3, 512, 640
    fine = torch.randn(B, 8, H, W, generator=g).cuda()
    coarse = torch.randn(B, 32, H // 2, W // 2, generator=g).cuda()
    wl, bl = torch.randn(32, 8, 1, 1, generator=g).cuda(), torch.randn(32, generator=g).cuda()
    ws, bs = (torch.randn(8, 32, 3, 3, generator=g) / 6).cuda(), torch.randn(8, generator=g).cuda()
    rgb = torch.rand(B, 3, H, W, generator=g).cuda()
    wsp, bt = convnet.pack_fpn_smooth_s(ws, bs, wl, bl, order=convnet.LookupRecords.EVEN_ODD)
    print("TIMES %.2f %.2f" % (timed(lambda: convnet.fpn_smooth_s(fine, coarse, wsp, bt, rgb=rgb)),
                               timed(lambda: convnet.fpn_smooth_s(fine, coarse, wsp, bt))))


def main():
    from boostmvsnerfs_amd import build
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in build.SOURCES if s != "fpn_s.hip"]
    print(f"{'flags':>5s}  {'build':28s} {'records':>10s} {'planar':>10s}   (us)")
    for fl, what in FLAGS:
        o, lib = f"/tmp/fpn_s_ab{fl}.o", f"/tmp/libbmv_fpn_s_ab{fl}.so"
        subprocess.check_call([build._hipcc(), *build.FLAGS, f"-DBMV_FPN_S_ABLATE={fl}", "-c", os.path.join(CSRC, "fpn_s.hip"), "-o", o])
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, o])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, BMV_LIB_PATH=lib),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("TIMES ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(1)
        a, b = line[0].split()[1:]
        print(f"{fl:5d}  {what:28s} {float(a):10.1f} {float(b):10.1f}", flush=True)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
