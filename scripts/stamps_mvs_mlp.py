"""Phase stamps of MVSNeRF's 6 x 128 MLP in its bf16 x 3 form (csrc/mvs.hip, bmv_mvs_mlp_fwd alone at one render launch of
BASELINE configs[3]): shader-clock cycles between the phase boundaries of every wave's third tile.  Builds a tuning library
with -DBMV_MVS_STAMPS under /tmp; the stamps drain every counter, so they serialise what the shipped kernel overlaps
(the sum is larger than the shipped kernel's tile period, printed last).
    python scripts/stamps_mvs_mlp.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "boostmvsnerfs_amd", "csrc")
LIB = "/tmp/libbmv_mvs_stamps.so"
NS = 52


def child():
    import numpy as np
    import torch
    from boostmvsnerfs_amd import _lib, ops
    torch.manual_seed(0)
    dims = {"pts_linears.0": (128, 63), "pts_linears.1": (128, 128), "pts_linears.2": (128, 128), "pts_linears.3": (128, 128),
            "pts_linears.4": (128, 128), "pts_linears.5": (128, 191), "pts_bias": (128, 20), "views_linears.0": (64, 131),
            "feature_linear": (128, 128), "alpha_linear": (1, 128), "rgb_linear": (3, 64)}
    w = {k: torch.randn(*s, device="cuda") / s[1] ** 0.5 for k, s in dims.items()}
    b = {k: torch.randn(s[0], device="cuda") * 0.1 for k, s in dims.items()}
    blob = ops.mvs_mlp_pack_weights(w, b)
    x = torch.randn(224 * 352 * 32, 86, device="cuda")
    _lib.set_tuning("BMV_MVS_SPLIT", 1)
    for _ in range(2):
        ops.mvs_mlp(x, blob)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(LIB)
    buf = (ctypes.c_float * (256 * 4 * NS))()
    assert lib.bmv_debug_fetch_mvs_stamps(buf) == 0
    s = np.frombuffer(buf, dtype=np.float32).reshape(-1, NS)
    s = s[s[:, 0] == 1.0]
    print(len(s), "waves; cycles of the shader clock (s_memtime), median over the waves")
    names = {1: "pts_bias: 40 fp32 MFMAs (weights resident in registers)"}
    for layer in range(6):
        o = 2 + 6 * layer
        names[o] = f"pts_linears.{layer}: operand split"
        names[o + 1] = "    accumulator load + chunk wait (A)"
        names[o + 2] = "    matrix chunk A"
        names[o + 3] = "    accumulator load + chunk wait (B)"
        names[o + 4] = "    matrix chunk B"
        names[o + 5] = "    bias * relu epilogue"
    names.update({38: "alpha head", 39: "feature_linear: operand split", 40: "    accumulator load + chunk wait (A)",
                  41: "    matrix chunk A", 42: "    accumulator load + chunk wait (B)", 43: "    matrix chunk B",
                  44: "views_linears.0: accumulator load + chunk wait", 45: "    operand split", 46: "    matrix chunk",
                  47: "rgb head"})
    prev = np.zeros(len(s))
    tot = {}
    for i in range(1, 48):
        d = s[:, i] - prev
        prev = s[:, i]
        print(f"  {names[i]:52s} {np.median(d):8.0f}")
        key = ("split" if "split" in names[i] else "matrix" if "matrix" in names[i] else "wait" if "wait" in names[i]
               else "epilogue" if "epilogue" in names[i] else "heads" if "head" in names[i] else "pts_bias")
        tot[key] = tot.get(key, 0) + float(np.median(d))
    print("  sums:", ", ".join(f"{k} {v:.0f}" for k, v in tot.items()), f"| all {float(np.median(s[:, 47])):.0f}")
    print(f"  tile period, start to start (input loads included): {float(np.median(s[:, 48])):.0f}")


def main():
    from boostmvsnerfs_amd import build
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in build.SOURCES if s != "mvs.hip"]
    o = "/tmp/mvs_stamps.o"
    subprocess.check_call([build._hipcc(), *build.FLAGS, "-DBMV_MVS_STAMPS", "-c", os.path.join(CSRC, "mvs.hip"), "-o", o])
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, o])
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, BMV_LIB_PATH=LIB))
    raise SystemExit(r.returncode)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
