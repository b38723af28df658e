"""MLP-only time of the ENeRF renderer (a11 without the gathers) at the sample count of BASELINE configs[1]:
how much of the fused render kernel is the MLP."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boostmvsnerfs_amd import ops  # noqa: E402
from boostmvsnerfs_amd.config import make_cfg, set_cfg  # noqa: E402
set_cfg(make_cfg("enerf_eval"))
from boostmvsnerfs_amd.networks.enerf.nerf import NeRF  # noqa: E402

P = 512 * 640 * 2
nerf = NeRF(feat_ch=8 + 3).cuda()
blob = nerf.packed_weights()
vox = torch.randn(P, 8, device="cuda")
img = torch.randn(P, 3, 8 + 3 + 4, device="cuda")
for _ in range(3):
    ops.nerf_mlp(vox, img, blob, 8)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ops.nerf_mlp(vox, img, blob, 8)
e.record()
torch.cuda.synchronize()
print(f"nerf_mlp on {P} samples: {s.elapsed_time(e) / 20 * 1e3:.1f} us")
