"""Whole-frame parity at the FULL sizes of BASELINE configs[0] / [1] and module-level golden tests of the convolution
engine.

* `test_whole_frame_matches_oracle`: `Network.forward` -- issued eagerly AND replayed as the HIP graph bench.py times --
  against `oracle.enerf.enerf_forward` (the CPU restatement of lib/networks/enerf/network.py:76-113) on the same
  weights and the same batch, at 256x320 / planes [32, 8] and 512x640 / planes [64, 8].  The oracle takes 2-3 s / ~7 s
  of host time for these frames (what bench.py's cpu_baseline leg runs), so the strongest check there is fits a test.
* `test_*_module_matches_reference`: FeatureNet, the two ENeRF regularisers and MVSNeRF's feature / regulariser stacks
  on the convolution engine against the activations the REFERENCE modules produced (tests/golden/*.npz `cap/...`).
"""
import pytest
import torch

from conftest import assert_close, load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _perturb(net, seed=1):
    """The golden fixtures' weight recipe (tests/golden/make_golden.py perturb_): non-zero biases, non-trivial BN
    statistics, peaky depth distributions."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in net.named_modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                m.weight.copy_(0.75 + 0.5 * torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
            elif isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d, nn.ConvTranspose3d)):
                if m.bias is not None:
                    m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                if "depth_conv" in name:
                    m.weight.mul_(40.0)
    return net


# (H, W, volume_planes): BASELINE configs[0] (the reference's CPU-runnable case) and configs[1] (the metric's)
FULL = {"config1_256x320": (256, 320, [32, 8]), "config2_512x640": (512, 640, [64, 8])}


@pytest.mark.parametrize("name", list(FULL))
def test_whole_frame_matches_oracle(name):
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.framegraph import FrameGraph
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from oracle import enerf as O      # the checker
    H, W, planes = FULL[name]
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = list(planes)
    set_cfg(cfg)
    torch.manual_seed(0)
    net = _perturb(Network().eval())
    batch = make_batch(H, W, n_views=3, seed=0)
    with torch.no_grad():
        want = O.enerf_forward({k: v.clone() for k, v in net.state_dict().items()}, clone_batch(batch), cfg)
    net = net.to(DEV)
    keys = ("rgb_level1", "depth_level1", "std_level1", "depth_mvs_level1")
    assert set(keys) <= set(want)
    with torch.no_grad():
        eager = net(clone_batch(batch, DEV))
    torch.cuda.synchronize()
    assert set(eager) == set(want)
    graph = FrameGraph(net, clone_batch(batch, DEV), cut=None)
    replay = graph.replay()
    torch.cuda.synchronize()
    for label, got in (("eager", eager), ("graph replay", replay)):
        for k in want:
            # the project bar: |d| <= 1e-3 |want| + 1e-3 rms(want); a handful of rays sit on a depth-distribution
            # mode switch of the 40x-sharpened logits (argmax-like softmax), where one ulp of the volume moves the ray's
            # sample range: budget 1e-4 of the entries, counted by assert_close
            assert_close(got[k], want[k], name=f"{name} {label} {k}", max_outlier_frac=1e-4)
        mse = float(((got["rgb_level1"].cpu() - want["rgb_level1"]) ** 2).mean())
        assert mse < 1e-8, f"{name} {label}: mse between renders {mse:.3e}"
    # the replay is the same arithmetic as the eager call
    for k in want:
        assert float((replay[k] - eager[k]).abs().max()) <= 1e-5 * float(eager[k].abs().max()) + 1e-7, k


# ------------------------------------------------------------------ convolution engine vs the reference's activations
@pytest.fixture(scope="module")
def enerf_fx():
    return load_fixture("enerf_tiny")


def _enerf_net(fx):
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [int(fx.t("cap/build_feature_volume#0.0").shape[2]),
                                          int(fx.t("cap/build_feature_volume#1.0").shape[2])]
    set_cfg(cfg)
    net = Network()
    net.load_state_dict(fx.group("sd"), strict=True)
    return net.to(DEV).eval()


def test_feature_net_module_matches_reference(enerf_fx):
    """lib/networks/enerf/feature_net.py:4-36 on the engine (fused first block, conv2.1 + toplayer, FPN + smooth0)
    vs the three maps the reference module produced for the fixture's source views."""
    net = _enerf_net(enerf_fx)
    x = enerf_fx.batch(DEV)["src_inps"][0]
    with torch.no_grad():
        maps = net.feature_net(x)
    for i, m in enumerate(maps):
        assert_close(m.contiguous(), enerf_fx.t(f"cap/feature_net#0.{i}"), rtol=1e-3, atol_scale=1e-4, name=f"feature_net level {i}")


@pytest.mark.parametrize("lvl", [0, 1])
def test_cost_reg_module_matches_reference(enerf_fx, lvl):
    """lib/networks/enerf/cost_reg_net.py:4-86 (MinCostRegNet at level 0, CostRegNet at level 1) on the engine, fed
    the variance volume the REFERENCE's build_feature_volume produced, vs the reference's (feature volume, depth
    logits); both output forms of the head (planar and voxel records)."""
    from boostmvsnerfs_amd import convnet
    net = _enerf_net(enerf_fx)
    reg = getattr(net, f"cost_reg_{lvl}")
    var = enerf_fx.t(f"cap/build_feature_volume#{lvl}.0", DEV)
    with torch.no_grad():
        feat, prob = reg(var)
        assert_close(feat, enerf_fx.t(f"cap/cost_reg_{lvl}#0.0"), rtol=1e-3, atol_scale=1e-4, name=f"cost_reg_{lvl} feature volume")
        assert_close(prob, enerf_fx.t(f"cap/cost_reg_{lvl}#0.1"), rtol=1e-3, atol_scale=1e-4, name=f"cost_reg_{lvl} depth logits")
        reg.volume_records = True
        try:
            rec, prob_r = reg(var)
        finally:
            reg.volume_records = False
    assert isinstance(rec, convnet.VolumeRecords)
    # record = [ch 0 2 4 6 | ch 1 3 5 7] of a voxel, (B,D,h,w,8)
    planar = rec.t[..., [0, 4, 1, 5, 2, 6, 3, 7]].permute(0, 4, 1, 2, 3)
    assert_close(planar.contiguous(), enerf_fx.t(f"cap/cost_reg_{lvl}#0.0"), rtol=1e-3, atol_scale=1e-4, name=f"cost_reg_{lvl} voxel records")
    assert_close(prob_r, enerf_fx.t(f"cap/cost_reg_{lvl}#0.1"), rtol=1e-3, atol_scale=1e-4, name=f"cost_reg_{lvl} logits (records form)")


def test_mvsnerf_modules_match_reference():
    """MVSNeRF's FeatureNet and CostRegNet (InPlaceABN stacks, lib/networks/mvsnerf/network.py:699-779) on the
    engine vs the reference's `feature` / `cost_reg_2` captures."""
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.mvsnerf.network import Network
    fx = load_fixture("mvsnerf_tiny")
    c = make_cfg("mvsnerf_eval")
    c.enerf.cas_config.num_samples = [int(x) for x in fx.raw["extra/num_samples"]]
    set_cfg(c)
    net = Network()
    net.load_state_dict(fx.group("sd"), strict=True)
    net = net.to(DEV).eval()
    b = fx.batch(DEV)
    with torch.no_grad():
        f = net.feature(b["all_src_inps"])                # (B,V,32,H/4,W/4); the plain network builds its volume from views 0..2
        want_f = fx.t("cap/feature#0")
        assert_close(f[:, :want_f.shape[1]].contiguous(), want_f, rtol=1e-3, atol_scale=1e-4, name="mvsnerf feature")
        vol = net.cost_reg_2(fx.t("cap/build_volume_costvar_img#0", DEV))
        vol = vol[0] if isinstance(vol, (tuple, list)) else vol
        assert_close(vol, fx.t("cap/cost_reg_2#0"), rtol=1e-3, atol_scale=1e-4, name="mvsnerf cost_reg_2")
