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
# entries of an output allowed outside the project bar at full size.  Measured on MI355X / ROCm 7.2 (round 4, the prints of
# this test): 0 for every output of both configs, eager and replayed; the ceiling is that plus a few entries for a ray on a
# mode switch under another evaluation order -- rounds 2-3 allowed 1e-4 of the entries (98 of the 512x640 rgb)
OUTLIER_CEILING = 6


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
            # the project bar: |d| <= 1e-3 |want| + 1e-3 rms(want).  A ray can sit on a depth-distribution mode switch of
            # the 40x-sharpened logits (argmax-like softmax), where one ulp of the volume moves its sample range; such
            # entries are COUNTED and held to a ceiling pinned to the measurement (OUTLIER_CEILING), not to a fraction
            g, w = got[k].detach().float().cpu(), want[k]
            bad = int(((g - w).abs() > 1e-3 * w.abs() + 1e-3 * float(w.pow(2).mean().sqrt())).sum())
            print(f"[whole frame] {name} {label} {k}: {bad} of {w.numel()} entries outside the bar")
            assert bad <= OUTLIER_CEILING, f"{name} {label} {k}: {bad} entries outside the bar (ceiling {OUTLIER_CEILING})"
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


# ------------------------------------------------------------------ BASELINE configs[2], [3] at size vs the oracle
def _flip_aware(out, want, masks, ref_masks, keys, tag, flip_ceiling, outlier_frac=0.0):
    """The viewport test of a14 is a step function: samples whose visibility differs from the oracle's are COUNTED (ceiling
    pinned to the measurement on MI355X), rays without such a sample are held to the project bar, the others may move."""
    diff = (masks.cpu() - ref_masks).abs() > 1e-6          # (1, K, n, Ns)
    per_volume = diff.sum((0, 2, 3)).tolist()
    print(f"[{tag}] visibility flips per volume: {per_volume} of {diff[0, 0].numel()} samples")
    assert max(per_volume) <= flip_ceiling, f"{tag}: {per_volume} visibility flips (ceiling {flip_ceiling})"
    flipped = diff.any(-1).any(1)[0]                       # (n,) rays with a flipped sample in any volume
    for k in keys:
        g, w = out[k].detach().float().cpu(), want[k]
        bad = ((g - w).abs() > 1e-3 * w.abs() + 1e-3 * float(w.pow(2).mean().sqrt()))
        print(f"[{tag}] {k}: {int(bad.sum())} of {bad.numel()} entries outside the bar ({int(flipped.sum())} rays carry a flipped sample)")
        if g.dim() >= 2 and g.shape[1] == flipped.shape[0]:
            assert_close(g[:, ~flipped], w[:, ~flipped], name=f"{tag} {k}", max_outlier_frac=outlier_frac)
        else:
            assert_close(g, w, name=f"{tag} {k}", max_outlier_frac=outlier_frac)


def test_config3_full_size_matches_oracle_on_a_ray_subset(tmp_path):
    """BASELINE configs[2]: enerf_ours at 480x736, N = 6 source views, K = 4 cost volumes, planes [64, 8]
    (lib/networks/boost_enerf/network.py:172-237).  The front end (FeatureNet on 6 views, 4 x 2 cost volumes and
    regularisers) is run IN FULL by the oracle; the per-ray part on every 8th ray -- the sample bench.py's cpu_baseline
    leg times (~7 s of host time).  Flip-aware: see _flip_aware."""
    import json
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.boost_enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from oracle import enerf as O      # the checker
    cfg = make_cfg("enerf_ours_eval")
    cfg.enerf.cas_config.volume_planes = [64, 8]
    cfg.enerf.cas_config.k_best = 4
    cfg.result_dir = str(tmp_path)
    set_cfg(cfg)
    sel = [0, 7, 13, 19]
    with open(tmp_path / "view_selection.json", "w") as f:
        json.dump({"synthetic_0": sel}, f)
    torch.manual_seed(0)
    net = _perturb(Network().eval())
    batch = make_batch(480, 736, n_views=6, seed=0)
    for i in range(cfg.enerf.cas_config.num):
        batch[f"rays_{i}"] = batch[f"rays_{i}"][:, ::8].contiguous()
    cap = {}
    with torch.no_grad():
        want = O.boost_enerf_forward({k: v.clone() for k, v in net.state_dict().items()}, clone_batch(batch), cfg, sel, capture=cap)
    net = net.to(DEV)
    net.capture = {}
    with torch.no_grad():
        out = net(clone_batch(batch, DEV))
    torch.cuda.synchronize()
    assert out["rgb_level1"].shape == (1, 480 * 736 // 8, 3)
    # measured on MI355X / ROCm 7.2 (this test's print): see the ceilings below
    # measured on MI355X / ROCm 7.2 (this test's prints): 0 flips in every volume, 0 entries outside the bar
    _flip_aware(out, want, net.capture["level1"][2], cap["masks_1"], ("rgb_level1", "depth_level1"), "config 3", flip_ceiling=2)
    for k in ("depth_mvs_level1", "std_level1"):
        assert_close(out[k], want[k], name=f"config 3 {k}")


def test_config4_full_size_matches_oracle_on_a_ray_subset(tmp_path):
    """BASELINE configs[3]: mvsnerf_ours at 224x352, 128 depth planes AND 128 samples per ray, K = 4
    (lib/networks/boost_mvsnerf/network.py:160-211): the K padded cost volumes and regularisers in full, every 64th ray
    through the sampler, the 6 x 128 MLP (157 k points) and the fusion."""
    import json
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.boost_mvsnerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from oracle import mvsnerf as M    # the checker
    cfg = make_cfg("mvsnerf_ours_eval")
    cfg.enerf.cas_config.num_samples = [128]
    cfg.enerf.cas_config.k_best = 4
    cfg.result_dir = str(tmp_path)
    set_cfg(cfg)
    sel = [0, 7, 13, 19]
    with open(tmp_path / "view_selection.json", "w") as f:
        json.dump({"synthetic_0": sel}, f)
    torch.manual_seed(0)
    net = _perturb(Network().eval())
    batch = make_batch(224, 352, n_views=6, seed=0, depth_ranges=True, render_scales=(1.0,))
    batch["rays_0"][..., 6], batch["rays_0"][..., 7] = 2.2, 7.5      # a real depth interval (the loaders' quirk 9 puts x, y there)
    batch["rays_0"] = batch["rays_0"][:, ::64].contiguous()
    cap = {}
    with torch.no_grad():
        want = M.boost_mvsnerf_forward({k: v.clone() for k, v in net.state_dict().items()}, clone_batch(batch), cfg, sel, capture=cap)
    net = net.to(DEV)
    net.capture = {}
    with torch.no_grad():
        out = net(clone_batch(batch, DEV))
    torch.cuda.synchronize()
    assert out["rgb_level0"].shape == (1, 224 * 352 // 64, 3)
    # measured: 0 flips of 157 696 samples per volume, 0 entries outside the bar
    _flip_aware(out, want, net.capture["masks"], cap["masks"], ("rgb_level0", "depth_level0"), "config 4", flip_ceiling=2)


def test_config5_full_size_gradients_match_oracle_on_a_ray_subset(tmp_path):
    """BASELINE configs[4]: the enerf_ours_ft fine-tune step at 480x736, N = 6, K = 4, both levels rendered
    (lib/train/trainers/trainer.py:44-63, lib/train/losses/enerf.py:7-56): loss and ALL 115 parameter gradients of the
    HIP forward + backward.  The front end (FeatureNet on 6 views, 4 x 2 cost volumes and regularisers, with their whole
    backward) runs in full on both sides, the per-ray part on every 16th ray of each level (the sample bench.py's
    cpu_baseline leg extrapolates from).

    At this size a weight gradient is an fp32 sum over ~1e6 voxels and the fp32 ORACLE itself is up to 10 x the tiny
    fixtures' 2e-3 bar away from its own float64 run (tests/tools/config5_grad_probe.py --fp64 ->
    profiles/r5/config5_grad_probe_fp64.txt: worst entry / tol 10.6 for the fp32 oracle, 7.9 for HIP; HIP the closer one
    on 101 of 115 tensors).  So the truth here is the oracle in FLOAT64, and the yardstick for every tensor is how far
    the fp32 oracle -- the reference's own arithmetic -- sits from it: the HIP gradient's relative L2 distance from the
    float64 gradient must not exceed 1.5 x the fp32 oracle's (+ 1e-3), and HIP must be the closer of the two on most
    tensors.  Both scatter forms: float atomics and the bit-reproducible fixed-point accumulation."""
    import json
    from boostmvsnerfs_amd import _lib
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.boost_enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from boostmvsnerfs_amd.train import NetworkWrapper
    from oracle import enerf as O      # the checker
    cfg = make_cfg("enerf_ours_ft")
    cc = cfg.enerf.cas_config
    cc.volume_planes = [64, 8]
    cc.k_best = 4
    cfg.result_dir = str(tmp_path)
    set_cfg(cfg)
    sel = [0, 7, 13, 19]
    with open(tmp_path / "view_selection.json", "w") as f:
        json.dump({"synthetic_0": sel}, f)
    torch.manual_seed(0)
    net = _perturb(Network().eval())
    batch = make_batch(480, 736, n_views=6, seed=0)
    g = torch.Generator().manual_seed(3)
    for i in range(cc.num):
        batch[f"rays_{i}"] = batch[f"rays_{i}"][:, ::16].contiguous()
        batch[f"rgb_{i}"] = torch.rand(1, batch[f"rays_{i}"].shape[1], 3, generator=g)

    def oracle(dtype):
        torch.set_default_dtype(dtype)
        try:
            leaves = {k: (v.detach().to(dtype) if v.is_floating_point() else v.clone()).requires_grad_(
                v.is_floating_point() and "running" not in k) for k, v in net.state_dict().items()}
            bb = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in clone_batch(batch).items()}
            out = O.boost_enerf_forward(leaves, bb, cfg, sel)
            loss = sum(cc.loss_weight[i] * ((out[f"rgb_level{i}"] - bb[f"rgb_{i}"]) ** 2).mean()
                       for i in range(cc.num) if f"rgb_level{i}" in out)
            loss.backward()
            return float(loss.detach()), {k: v.grad.double() for k, v in leaves.items() if v.requires_grad}
        finally:
            torch.set_default_dtype(torch.float32)
    loss32, g32 = oracle(torch.float32)
    loss64, g64 = oracle(torch.float64)
    assert len(g64) == 115
    gmax = max(float(v.abs().max()) for v in g64.values())

    def dist(x, k):
        return float((x - g64[k]).pow(2).sum().sqrt()) / (float(g64[k].pow(2).sum().sqrt()) + 1e-6 * gmax)
    net = net.to(DEV)
    bg = clone_batch(batch, DEV)
    before = _lib.get_tuning("BMV_DETERMINISTIC")
    try:
        for det in (0, 1):
            _lib.set_tuning("BMV_DETERMINISTIC", det)
            net.zero_grad(set_to_none=True)
            _, loss, _, _ = NetworkWrapper(net)(bg)
            loss.mean().backward()
            assert abs(float(loss) - loss64) <= 1e-5 * abs(loss64), (float(loss), loss64, loss32)
            closer, worst = 0, (0.0, None)
            for k, p in net.named_parameters():
                assert p.grad is not None, f"{k} received no gradient"
                d_hip, d_o32 = dist(p.grad.double().cpu(), k), dist(g32[k], k)
                closer += d_hip <= d_o32
                worst = max(worst, (d_hip / max(d_o32, 1e-12), k))
                assert d_hip <= 1.5 * d_o32 + 1e-3, (f"{k} (deterministic={det}): HIP is {d_hip:.3e} from the float64 gradient, "
                                                     f"the fp32 oracle {d_o32:.3e}")
            print(f"config 5 gradients (deterministic={det}): HIP closer to float64 than the fp32 oracle on {closer} of 115 "
                  f"tensors; worst ratio {worst[0]:.2f} ({worst[1]})")
            assert closer >= 70, closer
    finally:
        _lib.set_tuning("BMV_DETERMINISTIC", before)
