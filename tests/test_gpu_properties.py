"""Size-independent properties of the hot path at the FULL size of BASELINE configs[1]
(ENeRF, 512x640 target, 3 source views, planes [64, 8]) where the CPU oracle is too slow to be the
checker: symmetries of the variance sweep, bit-exact ray sharding, K = 1 fusion == plain compositing,
range invariants of depth regression / compositing, run-to-run determinism, and the convolution engine
against the torch modules it replaces (same device, same weights)."""
import pytest
import torch

from boostmvsnerfs_amd import switches

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, W = 512, 640


@pytest.fixture(scope="module")
def full():
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [64, 8]
    set_cfg(cfg)
    torch.manual_seed(0)
    net = Network().eval().to(DEV)
    batch = clone_batch(make_batch(H, W, n_views=3, seed=0), DEV)
    return cfg, net, batch


def _front(net, batch, cfg):
    """features, proj, hypotheses of both levels exactly as Network.forward builds them."""
    from boostmvsnerfs_amd import ops
    cc = cfg.enerf.cas_config
    with torch.no_grad():
        feats = net.forward_feat(batch["src_inps"])
        views = (batch["src_inps"], batch["src_exts"], batch["src_ixts"])
        st0 = net.level_front(0, feats["level_0"], views, batch, None)
        st1 = net.level_front(1, feats["level_1"], views, batch, st0)
        proj0 = ops.proj_mats(batch["src_exts"], batch["src_ixts"], batch["tar_ext"], batch["tar_ixt"],
                              cc.im_feat_scale[0], cc.volume_scale[0])
        proj1 = ops.proj_mats(batch["src_exts"], batch["src_ixts"], batch["tar_ext"], batch["tar_ixt"],
                              cc.im_feat_scale[1], cc.volume_scale[1])
    return feats, (st0, st1), (proj0, proj1)


@pytest.mark.parametrize("level", [0, 1])
def test_sweep_symmetries_full_size(full, level):
    from boostmvsnerfs_amd import ops
    cfg, net, batch = full
    feats, sts, projs = _front(net, batch, cfg)
    f = feats[f"level_{level}"].contiguous()            # reference layout (B,S,C,Hs,Ws)
    proj, dv = projs[level], sts[level].depth_values
    assert dv.shape[1] == cfg.enerf.cas_config.volume_planes[level]
    var = ops.sweep_variance(f, proj, dv)
    assert var.shape == (1, f.shape[2], dv.shape[1], dv.shape[2], dv.shape[3])
    assert bool(torch.isfinite(var).all())
    scale = float(f.pow(2).max())             # rounding of sum x^2/S - mean^2 scales with x^2, not with the variance
    assert float(var.min()) >= -1e-6 * scale                                    # a variance
    # (1) the variance over views does not depend on the order of the views
    perm = torch.tensor([2, 0, 1], device=DEV)
    var_p = ops.sweep_variance(f[:, perm].contiguous(), proj[:, perm].contiguous(), dv)
    assert float((var_p - var).abs().max()) <= 5e-6 * scale      # summation order over the views: a few ulp of x^2
    # (2) var(a x) = a^2 var(x)
    var_s = ops.sweep_variance(f * 3.0, proj, dv)
    assert float((var_s - 9.0 * var).abs().max()) <= 1e-5 * 9.0 * scale
    # (3) the channel-last fast path and the reference-layout direct-gather kernel agree
    #     (v_rcp_f32 vs IEEE division moves a tap coordinate by an ulp: the project's 1e-3 bar applies)
    from conftest import assert_close
    assert_close(ops.sweep_variance(f, proj, dv, algo=1), var, name="direct-gather vs channel-last sweep")
    # (4) identical views -> zero variance wherever every view sees the voxel; with ONE view repeated the
    #     variance is exactly sum x^2/S - (sum x/S)^2 = 0 up to rounding everywhere
    same_f = f[:, :1].expand(-1, 3, -1, -1, -1).contiguous()
    same_p = proj[:, :1].expand(-1, 3, -1, -1).contiguous()
    var_0 = ops.sweep_variance(same_f, same_p, dv)
    assert float(var_0.abs().max()) <= 1e-5 * float(f.pow(2).max())


def test_views_by_index_equals_gathered_views_full_size(full):
    """bmv_sweep_variance_views_fwd / view_ids of the render kernel (boost path: every cost volume picks 3 of the N
    source views) == the same kernels on gathered copies of those views, bit for bit."""
    from boostmvsnerfs_amd import ops
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    cfg, net, _ = full
    cc = cfg.enerf.cas_config
    batch = clone_batch(make_batch(H, W, n_views=5, seed=1), DEV)
    ids = torch.tensor([[3, 0, 4]], device=DEV)
    ids32 = ids.to(torch.int32)
    bi = torch.arange(1, device=DEV)[:, None]
    with torch.no_grad():
        feats = net.forward_feat(batch["all_src_inps"])
        exts, ixts = batch["all_src_exts"][bi, ids], batch["all_src_ixts"][bi, ids]
        picked = (batch["all_src_inps"][bi, ids], exts, ixts)
        by_index = (batch["all_src_inps"], exts, ixts)
        f1 = feats["level_1"]
        assert isinstance(f1, ops.QuadFeats)                 # the inference sweep's layout, all N views
        st_a = st_b = None
        for i in range(cc.num):
            fa = ops.QuadFeats(feats[f"level_{i}"].data[bi, ids].contiguous())      # gathered copies of the 3 views
            st_a = net.level_front(i, fa, picked, batch, st_a)
            st_b = net.level_front(i, feats[f"level_{i}"], by_index, batch, st_b, view_ids=ids32)
            assert torch.equal(st_a.depth, st_b.depth) and torch.equal(st_a.feature_volume, st_b.feature_volume)
        out_a = net.render_level(1, st_a, feats["level_2"][bi, ids], picked, batch)
        out_b = net.render_level(1, st_b, feats["level_2"], by_index, batch, view_ids=ids32)
    for a, b in zip(out_a, out_b):
        assert torch.equal(a, b)


def test_batch_of_two_equals_two_batches_of_one():
    """Every kernel takes the batch dimension in its grid: a B = 2 forward is bit-identical to the two B = 1 forwards
    (different cameras / rays per item)."""
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from boostmvsnerfs_amd.config import get_cfg
    prev = get_cfg()
    cfg = make_cfg("enerf_pretrain")                      # both levels rendered
    cfg.enerf.cas_config.volume_planes = [16, 8]
    set_cfg(cfg)
    try:
        torch.manual_seed(0)
        net = Network().eval().to(DEV)
        b2 = clone_batch(make_batch(64, 96, n_views=3, seed=0, B=2), DEV)
        with torch.no_grad():
            both = net(b2)
            singles = []
            for i in range(2):
                bi = {k: (v[i:i + 1].contiguous() if torch.is_tensor(v) else v) for k, v in b2.items()}
                singles.append(net(bi))
    finally:
        set_cfg(prev)                                     # the module-scope `full` fixture reads the global cfg
    assert {"rgb_level0", "rgb_level1"} <= set(both)
    for k, v in both.items():
        assert torch.equal(v, torch.cat([s[k] for s in singles], 0)), k


def test_depth_regression_ranges_full_size(full):
    cfg, net, batch = full
    _, (st0, st1), _ = _front(net, batch, cfg)
    for st, inv in ((st0, True), (st1, False)):
        dv = st.depth_values
        vals = torch.reciprocal(dv.clamp_min(1e-6)) if inv else dv
        lo, hi = vals.amin(1), vals.amax(1)
        eps = 1e-5 * hi.abs()
        assert bool(((st.depth >= lo - eps) & (st.depth <= hi + eps)).all())    # an expectation over the hypotheses
        assert bool((st.std >= 0).all()) and bool((st.std <= (hi - lo) + eps).all())


def test_ray_sharding_bit_exact_full_size(full):
    cfg, net, batch = full
    N = H * W
    with torch.no_grad():
        whole = net(batch)
        parts = []
        for rng in ((0, N // 3), (N // 3, N // 3 + 100_001), (N // 3 + 100_001, N)):   # ragged, not wave-aligned
            net.ray_range = rng
            parts.append(net(batch))
        net.ray_range = None
    for key in ("rgb_level1", "depth_level1", "weights_level1"):
        got = torch.cat([p[key] for p in parts], 1)
        assert got.shape == whole[key].shape
        assert torch.equal(got, whole[key]), key


def test_forward_invariants_and_determinism_full_size(full):
    cfg, net, batch = full
    with torch.no_grad():
        a = net(batch)
        b = net(batch)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k} differs between two runs on the same input"
        assert bool(torch.isfinite(a[k]).all()), k
    rgb, w, depth = a["rgb_level1"], a["weights_level1"], a["depth_level1"]
    assert rgb.shape == (1, H * W, 3) and depth.shape == (1, H * W) and w.shape == (1, H * W, 2)
    # colours are convex combinations of source colours in [0,1] weighted by compositing weights that sum to <= 1
    assert float(rgb.min()) >= -1e-6 and float(rgb.max()) <= 1 + 1e-5
    assert float(w.min()) >= 0 and float(w.sum(-1).max()) <= 1 + 1e-5
    nf = batch["near_far"].view(-1)
    assert float(depth.min()) >= float(nf[0]) * (1 - 1e-4) and float(depth.max()) <= float(nf[1]) * (1 + 1e-4)


def test_blend_with_one_volume_is_plain_compositing_full_size():
    from boostmvsnerfs_amd import ops
    g = torch.Generator().manual_seed(3)
    N, Ns = H * W, 2
    raw = torch.rand(1, N, Ns, 4, generator=g).to(DEV)
    raw[..., 3] *= 4.0
    z = (torch.rand(1, N, Ns, generator=g).sort(-1).values * 6 + 2).to(DEV)
    rgb_c, depth_c, w_c = ops.composite(raw, z, False)
    ones = torch.ones(1, 1, N, Ns, device=DEV)
    rgb_b, depth_b, w_b = ops.blend(raw[:, None].contiguous(), ones, z[:, None].contiguous(), normalise=True)
    assert float((rgb_b - rgb_c).abs().max()) <= 1e-6
    assert float((w_b - w_c).abs().max()) <= 1e-6
    assert float((depth_b - depth_c).abs().max()) <= 1e-5 * float(depth_c.abs().max())
    # two identical volumes with masks (1/2, 1/2) fuse to the same picture
    raws2 = raw[:, None].expand(-1, 2, -1, -1, -1).contiguous()
    rgb_2, _, _ = ops.blend(raws2, ones.expand(-1, 2, -1, -1).contiguous(), z[:, None].expand(-1, 2, -1, -1).contiguous(),
                            normalise=True)
    assert float((rgb_2 - rgb_c).abs().max()) <= 1e-5


def test_conv_engine_matches_torch_modules_full_size(full, monkeypatch):
    """FeatureNet on the 3 source views at 512x640 and both regularisers on the real variance volumes:
    the HIP convolution engine against the torch/MIOpen modules with the same parameters."""
    cfg, net, batch = full
    feats, (st0, st1), projs = _front(net, batch, cfg)
    from boostmvsnerfs_amd import ops
    var0 = ops.sweep_variance(feats["level_0"], projs[0], st0.depth_values)
    var1 = ops.sweep_variance(feats["level_1"], projs[1], st1.depth_values)
    x = batch["src_inps"].reshape(3, 3, H, W)
    with torch.no_grad():
        got_f = [t.contiguous() for t in net.feature_net(x)]
        got_0 = net.cost_reg_0(var0)
        got_1 = net.cost_reg_1(var1)
        monkeypatch.setitem(switches.VALUES, "BMV_CNN", "torch")
        want_f = net.feature_net(x)
        want_0 = net.cost_reg_0(var0)
        want_1 = net.cost_reg_1(var1)
    for got, want in list(zip(got_f, want_f)) + list(zip(got_0, want_0)) + list(zip(got_1, want_1)):
        assert got.shape == want.shape
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * scale
