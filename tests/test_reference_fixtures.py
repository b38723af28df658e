"""CPU checks of the host logic against fixtures generated from the reference itself (tests/golden/make_golden.py
`rays`, `cfg_dumps`, `adam_step`): the synthetic ray builder vs lib/datasets/enerf_utils.py:25-71, the configuration
presets vs the cfg the reference resolves for the five BASELINE configs (lib/config/config.py:170-188), and the
learning-rate schedule / optimiser settings vs lib/train/optimizer.py + lib/utils/optimizer/lr_scheduler.py:66-75."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_fixture

HERE = os.path.dirname(os.path.abspath(__file__))


def test_synthetic_ray_builder_matches_reference_rays():
    from boostmvsnerfs_amd.synthetic import make_rays
    fx = np.load(os.path.join(HERE, "golden", "rays_tiny.npz"))
    for c in range(2):
        H, W = (int(v) for v in fx[f"in/hw_{c}"])
        ext, ixt = fx[f"in/tar_ext_{c}"].astype(np.float64), fx[f"in/tar_ixt_{c}"].astype(np.float64)
        for level in range(2):
            scale = float(fx[f"extra/scale_{level}"])
            want = fx[f"out/rays_{c}_{level}"]
            got = make_rays(ext, ixt, H, W, scale)
            assert got.shape == want.shape == (int(round(H * scale)) * int(round(W * scale)), 8)
            assert np.array_equal(got[:, 6:], want[:, 6:])                       # pixel coordinates
            # the reference inverts K and the pose in float32, the builder in float64: a few float32 ulp
            assert float(np.abs(got - want).max()) <= 2e-6 * float(np.abs(want).max())


CFG_KEYS = ["num", "depth_inv", "volume_scale", "volume_planes", "im_feat_scale", "im_ibr_scale", "render_scale",
            "render_im_feat_level", "nerf_model_feat_ch", "num_samples", "render_if", "loss_weight"]
MVS_KEYS = ["num", "depth_inv", "render_scale", "num_samples", "render_if"]          # what the MVSNeRF modules read


@pytest.mark.parametrize("name,preset,opts,keys", [
    ("config1_enerf_256x320_32planes", "enerf_eval", ["enerf.cas_config.volume_planes", "[32, 8]"], CFG_KEYS),
    ("config2_enerf_512x640_64planes", "enerf_eval", [], CFG_KEYS),
    ("config3_enerf_ours_grass", "enerf_ours_ft", [], CFG_KEYS + ["k_best"]),
    ("config4_mvsnerf_ours_128", "mvsnerf_ours_eval", ["enerf.cas_config.num_samples", "[128]"], MVS_KEYS + ["k_best"]),
    ("config5_enerf_ours_ft_grass", "enerf_ours_ft", [], CFG_KEYS + ["k_best"]),
])
def test_presets_match_the_cfg_the_reference_resolves(name, preset, opts, keys):
    from boostmvsnerfs_amd.config import make_cfg
    want = json.load(open(os.path.join(HERE, "golden", "cfg_dumps.json")))[name]
    c = make_cfg(preset, opts=opts)
    for k in keys:
        assert c.enerf.cas_config[k] == want["cas_config"][k], (name, k, c.enerf.cas_config[k], want["cas_config"][k])
    for k, v in want["enerf"].items():
        assert c.enerf[k] == v, (name, k)
    assert c.network_module == want["network_module"]


def test_optimizer_and_lr_schedule_match_reference():
    from boostmvsnerfs_amd.train import make_lr_scheduler, make_optimizer
    fx = load_fixture("enerf_tiny_adam_step")
    net = torch.nn.Linear(3, 2)
    opt = make_optimizer(net)
    assert len(opt.param_groups) == 2                                   # one group per parameter, as the reference
    for g in opt.param_groups:
        assert g["lr"] == float(fx.raw["extra/lr0"]) and g["eps"] == float(fx.raw["extra/eps"]) and g["weight_decay"] == 0.0
    sched = make_lr_scheduler(opt)
    lrs = []
    for _ in range(101):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert np.allclose(np.asarray(lrs), fx.raw["extra/lr_by_epoch"], rtol=1e-12, atol=0)
