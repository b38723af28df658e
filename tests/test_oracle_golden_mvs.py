"""Pin oracle/mvsnerf.py against golden vectors produced by the reference's MVSNeRF /
boost_mvsnerf networks (tests/golden/make_golden.py mvsnerf|boost_mvsnerf).  CPU only."""
import pytest
import torch

from conftest import assert_close, load_fixture
from oracle import enerf as E
from oracle import mvsnerf as M


@pytest.fixture(scope="module")
def fx():
    return load_fixture("mvsnerf_tiny")


@pytest.fixture(scope="module")
def bfx():
    return load_fixture("boost_mvsnerf_tiny")


def mvs_cfg(f, preset="mvsnerf_eval"):
    from boostmvsnerfs_amd.config import make_cfg
    c = make_cfg(preset)
    c.enerf.cas_config.num_samples = [int(x) for x in f.raw["extra/num_samples"]]
    if "extra/k_best" in f.raw:
        c.enerf.cas_config.k_best = len(f.raw["extra/k_best"])
    return c


def chunks(f, name, dim=0):
    keys = sorted((k for k in f.raw if k.startswith(f"cap/{name}#")), key=lambda k: int(k.split("#")[1].split(".")[0]))
    return torch.cat([f.t(k) for k in keys], dim)


def test_cnns(fx):
    sd, b = fx.group("sd"), fx.batch()
    assert_close(M.feature_net(sd, b["all_src_inps"]), fx.t("cap/feature#0"), rtol=1e-4, atol_scale=1e-5, name="feature")
    assert_close(M.cost_reg(sd, fx.t("cap/build_volume_costvar_img#0")), fx.t("cap/cost_reg_2#0"), rtol=1e-4,
                 atol_scale=1e-5, name="cost_reg_2")


def test_proj_and_cost_volume(fx):
    b = fx.batch()
    P = M.proj_mats(b["all_src_exts"], b["all_src_ixts"])
    assert_close(P, fx.t("cap/get_proj_mats#0"), rtol=1e-4, atol_scale=1e-5, name="proj")
    dv, near, far = M.depth_planes(b["depth_ranges"], 8)
    vol = M.cost_volume(b["all_src_inps"], fx.t("cap/feature#0"), fx.t("cap/get_proj_mats#0"), dv[None])
    assert_close(vol, fx.t("cap/build_volume_costvar_img#0"), name="cost volume")


def test_point_inputs_and_mlp(fx):
    sd, b = fx.group("sd"), fx.batch()
    xyz, z = M.ray_march(b["rays_0"], 8)
    assert_close(xyz, fx.t("cap/ray_marcher#0.0"), rtol=1e-5, atol_scale=1e-6, name="xyz")
    assert_close(z, fx.t("cap/ray_marcher#0.1"), rtol=1e-5, atol_scale=1e-6, name="z")
    volume = fx.t("cap/cost_reg_2#0").reshape(1, 8, 8, 64, 72)
    _, near, far = M.depth_planes(b["depth_ranges"], 8)
    cap = {}
    x, _, _ = M.point_inputs(b["rays_0"], volume, b["all_src_inps"], b["all_src_exts"], b["all_src_ixts"], near, far, 8, cap)
    assert_close(cap["ndc"], chunks(fx, "get_ndc_coordinate"), rtol=1e-4, atol_scale=1e-5, name="ndc")
    assert_close(cap["angle"], chunks(fx, "gen_dir_feature", 1)[0], rtol=1e-5, atol_scale=1e-6, name="angle")
    assert_close(cap["feat"], chunks(fx, "gen_pts_feats"), name="pts feats")
    assert_close(x, chunks(fx, "run_network_mvs"), rtol=1e-3, atol_scale=2e-3, name="mlp input")  # sin/cos of 512*x
    raw = M.renderer_mlp(sd, chunks(fx, "run_network_mvs"))
    assert_close(raw, chunks(fx, "nerf"), name="mlp")


def test_mvsnerf_forward(fx):
    out = M.mvsnerf_forward(fx.group("sd"), fx.batch(), mvs_cfg(fx))
    want = fx.group("out")
    assert set(out) == set(want)
    for k in want:
        assert_close(out[k], want[k], name=k)


def test_boost_mvsnerf(fx, bfx):
    sd = fx.group("sd")                      # same seed + perturbation
    cfg = mvs_cfg(bfx, "mvsnerf_ours_eval")
    b = bfx.batch()
    assert M.view_selection(b, cfg) == {"synthetic_0": [int(k) for k in bfx.raw["extra/k_best"]]}
    trip = E.view_triplets(5, 3)
    for i in (0, 5, 9):
        assert_close(M.triplet_visibility(b, cfg, trip[i]), bfx.t(f"cap/sel/calc_mask#{i}.mask_level0"), rtol=1e-4,
                     atol_scale=1e-5, name=f"vis{i}")
    out = M.boost_mvsnerf_forward(sd, b, cfg, [int(k) for k in bfx.raw["extra/k_best"]])
    want = bfx.group("out")
    for k in want:
        assert_close(out[k], want[k], name=k, max_outlier_frac=2e-3)
