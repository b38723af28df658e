"""Build libbmv.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m boostmvsnerfs_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbmv.so")
SOURCES = ["sweep.hip", "sweep_win.hip", "sweep_quad.hip", "tuning.hip", "sample.hip", "render.hip", "mvs.hip", "mvs_mlp_train.hip", "backward.hip", "sweep_bwd_cl.hip", "mlp_bwd.hip", "conv.hip", "conv_c4.hip", "conv_c4s.hip", "fpn_s.hip", "conv2d_s.hip", "conv_split.hip", "conv_wgrad.hip", "bn.hip", "rays.hip", "timing.hip"]
HEADERS = ["bmv_common.hpp", "scatter.hpp", "sweep_util.hpp", "render_geom.hpp", "mlp.hpp", os.path.join("..", "..", "include", "bmv.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
EXTRA_FLAGS = {}    # per-file flags
# hipcc's SLP vectoriser turns the blend into v_pk_fma_f32 with every weight held as a register PAIR: slower per FMA on
# gfx950 (measured round 2) and 20 registers over the 5-waves-per-SIMD budget
EXTRA_FLAGS["sweep_quad.hip"] = ["-fno-slp-vectorize"]
if os.environ.get("BMV_QUAD_DEFS"):
    EXTRA_FLAGS["sweep_quad.hip"] += os.environ["BMV_QUAD_DEFS"].split()
if os.environ.get("BMV_RENDER_DEFS"):   # e.g. "-DBMV_RENDER_STAMPS" for scripts/stamps_render.py
    EXTRA_FLAGS["render.hip"] = os.environ["BMV_RENDER_DEFS"].split()
if os.environ.get("BMV_CONV_DEFS"):   # e.g. "-DBMV_CONV_WPE_TUNED=0": the convolution kernels at the allocator's own occupancy
    EXTRA_FLAGS["conv.hip"] = os.environ["BMV_CONV_DEFS"].split()
if os.environ.get("BMV_BWD_DEFS"):   # ablation builds of the scatter kernels (scripts/bench_sweep_bwd.py)
    EXTRA_FLAGS["backward.hip"] = os.environ["BMV_BWD_DEFS"].split()
if os.environ.get("BMV_MVS_DEFS"):
    EXTRA_FLAGS["mvs.hip"] = os.environ["BMV_MVS_DEFS"].split()
if os.environ.get("BMV_C4_DEFS"):   # ablation builds of the 4-row-block convolutions (scripts/ablate_conv_c4.py)
    EXTRA_FLAGS["conv_c4.hip"] = os.environ["BMV_C4_DEFS"].split()
if os.environ.get("BMV_C4S_DEFS"):  # ablation builds of the bf16 x 3 first layers / heads (scripts/ablate_conv_c4s.py)
    EXTRA_FLAGS["conv_c4s.hip"] = os.environ["BMV_C4S_DEFS"].split()
if os.environ.get("BMV_C2S_DEFS"):    # ablation builds of the bf16 x 3 encoder convolutions (scripts/ablate_conv2d_s.py)
    EXTRA_FLAGS["conv2d_s.hip"] = os.environ["BMV_C2S_DEFS"].split()
if os.environ.get("BMV_FPN_S_DEFS"):  # ablation builds of the bf16 x 3 top-down + smooth0 kernel (scripts/ablate_fpn_s.py)
    EXTRA_FLAGS["fpn_s.hip"] = os.environ["BMV_FPN_S_DEFS"].split()
if os.environ.get("BMV_WIN_DEFS"):   # kernel-tuning builds of the windowed sweep, e.g. "-DBMV_WIN_WPE=5 -DBMV_WIN_TAPBUF=1"
    EXTRA_FLAGS["sweep_win.hip"] = os.environ["BMV_WIN_DEFS"].split()


def _hipcc():
    return os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
