"""Where the time of MVSNeRF's 6 x 128 MLP in its bf16 x 3 form (csrc/mvs.hip) goes: ablation BUILDS
(-DBMV_MVS_ABLATE=n: results are wrong, timing only) of bmv_mvs_mlp_fwd at one render launch of BASELINE configs[3]
(224 x 352 x 32 points); every variant compiled to its own library under /tmp.
    python scripts/ablate_mvs_mlp.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "boostmvsnerfs_amd", "csrc")
FLAGS = [(0, "full kernel"), (1, "no operand split"), (2, "no bias * relu epilogue"), (4, "no chunk barrier / wait"),
         (8, "no matrix instructions"), (1 | 2, "no split, no epilogue"), (1 | 2 | 4, "matrix + LDS reads + heads"),
         (1 | 2 | 8, "LDS reads + barriers + heads"), (1 | 2 | 4 | 8, "skeleton"), (16, "no LDS reads of the A pieces"),
         (32, "accumulators start at zero"), (1 | 2 | 16 | 32, "matrix + barriers + heads + input"),
         (1 | 2 | 8 | 16, "skeleton without LDS reads"), (1 | 2 | 8 | 16 | 32, "... and accumulators from zero")]


def main():
    from boostmvsnerfs_amd import build
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in build.SOURCES if s != "mvs.hip"]
    print(f"{'flags':>5s}  {'build':32s} per tile and wave, BMV_MVS_SPLIT=1")
    for fl, what in FLAGS:
        o, lib = f"/tmp/mvs_ab{fl}.o", f"/tmp/libbmv_mvs_ab{fl}.so"
        subprocess.check_call([build._hipcc(), *build.FLAGS, f"-DBMV_MVS_ABLATE={fl}", "-c", os.path.join(CSRC, "mvs.hip"), "-o", o])
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, o])
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe_mvs_mlp.py")],
                           env=dict(os.environ, BMV_LIB_PATH=lib), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("BMV_MVS_SPLIT=1")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(1)
        print(f"{fl:5d}  {what:32s} {line[0].split(':', 1)[1].strip()}", flush=True)


if __name__ == "__main__":
    main()
