"""MVSNeRF's 6 x 128 MLP alone (bmv_mvs_mlp_fwd: rows of 86 inputs, no gathers) at the point count of one render
launch of BASELINE configs[3]: cycles per 32-sample tile and wave, for BMV_MVS_SPLIT=1 / 0.  Compared with the fused
render kernel's time per launch this says how much of that kernel is the MLP and how much the exposed gathers."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import _lib, ops  # noqa: E402

torch.manual_seed(0)
DEV = "cuda"
dims = {"pts_linears.0": (128, 63), "pts_linears.1": (128, 128), "pts_linears.2": (128, 128), "pts_linears.3": (128, 128),
        "pts_linears.4": (128, 128), "pts_linears.5": (128, 191), "pts_bias": (128, 20), "views_linears.0": (64, 131),
        "feature_linear": (128, 128), "alpha_linear": (1, 128), "rgb_linear": (3, 64)}
w = {k: torch.randn(*s, device=DEV) / s[1] ** 0.5 for k, s in dims.items()}
b = {k: torch.randn(s[0], device=DEV) * 0.1 for k, s in dims.items()}
blob = ops.mvs_mlp_pack_weights(w, b)
P = int(os.environ.get("POINTS", 224 * 352 * 32))
x = torch.randn(P, 86, device=DEV)
tiles_per_wave = (P + 31) // 32 / (256 * 4)
for split in (1, 0):
    _lib.set_tuning("BMV_MVS_SPLIT", split)
    for _ in range(2):
        ops.mvs_mlp(x, blob)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.mvs_mlp(x, blob)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"BMV_MVS_SPLIT={split}: {P} points {ms:.3f} ms, {ms * 1e3 / tiles_per_wave:.2f} us per tile and wave "
          f"({ms * 1e3 / tiles_per_wave * 2.4e3:.0f} cycles at 2.4 GHz), {P / ms / 1e6:.2f} Gpt/s")
