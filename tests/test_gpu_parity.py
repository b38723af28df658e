"""GPU parity: every HIP entry point (through the C ABI / ctypes) against the
golden vectors generated from the reference, function by function, then the
whole network.  Tolerance: 1e-3 relative (BASELINE.json north_star) with an
absolute floor of 1e-3 * rms(reference tensor) -- see conftest.assert_close."""
import pytest
import torch

from conftest import assert_close, tiny_cfg

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no fallback path exists)")
    from boostmvsnerfs_amd import ops as o
    return o


def g(fx, key):
    return fx.t(key, DEV)


def test_proj_mats(ops, enerf_fx):
    b = enerf_fx.batch(DEV)
    c = tiny_cfg(enerf_fx).enerf.cas_config
    for lvl in range(2):
        P = ops.proj_mats(b["src_exts"], b["src_ixts"], b["tar_ext"], b["tar_ixt"], c.im_feat_scale[lvl], c.volume_scale[lvl])
        assert_close(P, enerf_fx.t(f"cap/get_proj_mats#{lvl}"), rtol=1e-4, atol_scale=1e-5, name=f"proj{lvl}")


def test_non_contiguous_arguments_outlive_the_launch(ops, enerf_fx):
    """Several non-contiguous arguments in ONE call: each gets a contiguous copy, and every copy must still be intact
    when the kernel runs (the allocator used to hand the first copy's block to the second one)."""
    b = enerf_fx.batch(DEV)
    K = 3
    ext, ixt = b["src_exts"].expand(K, -1, -1, -1), b["src_ixts"].expand(K, -1, -1, -1)
    want = ops.proj_mats(b["src_exts"], b["src_ixts"], b["tar_ext"], b["tar_ixt"], 0.25, 0.125)
    for _ in range(3):
        got = ops.proj_mats(ext, ixt, b["tar_ext"].expand(K, -1, -1), b["tar_ixt"].expand(K, -1, -1), 0.25, 0.125)
        for k in range(K):
            assert torch.equal(got[k], want[0])


def test_depth_values(ops, enerf_fx):
    b = enerf_fx.batch(DEV)
    c = tiny_cfg(enerf_fx).enerf.cas_config
    dv0, nf0 = ops.depth_values_uniform(b["near_far"], c.volume_planes[0], 8, 12, True)
    assert_close(dv0, enerf_fx.t("cap/get_depth_values#0.0"), rtol=1e-5, atol_scale=0, name="dv0")
    assert_close(nf0, enerf_fx.t("cap/get_depth_values#0.1"), rtol=1e-5, atol_scale=0, name="nf0")
    dv1, nf1 = ops.depth_values_cascade(g(enerf_fx, "cap/depth_regression#0.0"), g(enerf_fx, "cap/depth_regression#0.1"),
                                        g(enerf_fx, "cap/get_depth_values#0.1"), 32, 48, c.volume_planes[1])
    assert_close(dv1, enerf_fx.t("cap/get_depth_values#1.0"), rtol=1e-5, atol_scale=0, name="dv1")
    assert_close(nf1, enerf_fx.t("cap/get_depth_values#1.1"), rtol=1e-5, atol_scale=0, name="nf1")


@pytest.mark.parametrize("algo", [1, 0])
def test_warp_and_sweep(ops, enerf_fx, algo):
    feats = {0: g(enerf_fx, "cap/feature_net#0.0")[None], 1: g(enerf_fx, "cap/feature_net#0.1")[None]}
    for lvl in range(2):
        P = g(enerf_fx, f"cap/get_proj_mats#{lvl}")
        dv = g(enerf_fx, f"cap/get_depth_values#{lvl}.0")
        w, grid = ops.homo_warp(feats[lvl][:, 1].contiguous(), P[:, 1].contiguous(), dv)
        call = 1 + 3 * lvl
        assert_close(grid, enerf_fx.t(f"cap/homo_warp#{call}.1"), rtol=1e-5, atol_scale=1e-5, name=f"grid{lvl}")
        assert_close(w, enerf_fx.t(f"cap/homo_warp#{call}.0"), name=f"warp{lvl}")
        var = ops.sweep_variance(feats[lvl], P, dv, algo=algo)
        assert_close(var, enerf_fx.t(f"cap/build_feature_volume#{lvl}.0"), name=f"var{lvl}")


def test_frame_setup_equals_the_separate_launches(ops, enerf_fx):
    """bmv_frame_setup = proj_mats of both levels + depth_values_uniform of level 0 in one launch, bit for bit."""
    b = enerf_fx.batch(DEV)
    c = tiny_cfg(enerf_fx).enerf.cas_config
    H, W = b["src_inps"].shape[-2:]
    h, w = int(H * c.volume_scale[0]), int(W * c.volume_scale[0])
    proj, (dv, nf) = ops.frame_setup(b["src_exts"], b["src_ixts"], b["tar_ext"], b["tar_ixt"],
                                     [c.im_feat_scale[i] for i in range(2)], [c.volume_scale[i] for i in range(2)],
                                     b["near_far"], c.volume_planes[0], h, w, c.depth_inv[0])
    for lvl in range(2):
        want = ops.proj_mats(b["src_exts"], b["src_ixts"], b["tar_ext"], b["tar_ixt"], c.im_feat_scale[lvl], c.volume_scale[lvl])
        assert torch.equal(proj[lvl], want), lvl
    dv2, nf2 = ops.depth_values_uniform(b["near_far"], c.volume_planes[0], h, w, c.depth_inv[0])
    assert torch.equal(dv, dv2) and torch.equal(nf, nf2)


def test_depth_regress(ops, enerf_fx):
    for lvl, inv in ((0, True), (1, False)):
        d, s = ops.depth_regress(g(enerf_fx, f"cap/cost_reg_{lvl}#0.1"), g(enerf_fx, f"cap/get_depth_values#{lvl}.0"), inv)
        assert_close(d, enerf_fx.t(f"cap/depth_regression#{lvl}.0"), rtol=1e-5, atol_scale=0, name=f"depth{lvl}")
        assert_close(s, enerf_fx.t(f"cap/depth_regression#{lvl}.1"), rtol=1e-4, atol_scale=1e-5, name=f"std{lvl}")


@pytest.mark.parametrize("D,h,w,inv", [(64, 64, 80, True), (64, 7, 9, True), (64, 5, 13, False), (32, 6, 10, True),
                                        (12, 5, 7, False)])
def test_depth_regress_all_plane_counts(ops, D, h, w, inv):
    """a5 at the plane counts of the shipped configs: 64 planes on a small map take the four-lanes-per-pixel kernel
    (ragged pixel counts: the last 16-pixel group is partial), 32 / 8 the one-thread kernels, 12 the generic loop."""
    from oracle import enerf as O
    torch.manual_seed(D + h)
    prob = torch.randn(2, D, h, w) * 3
    prob[0, :, 0, 0] = -40.0                       # a pixel whose logits are all equal and tiny
    prob[1, D // 2, 1, 1] = 60.0                   # a one-hot pixel
    dv = torch.rand(2, D, h, w) * 3 + 0.5
    want_d, want_s = O.depth_regress(prob, dv, inv)
    d, s = ops.depth_regress(prob.to(DEV), dv.to(DEV), inv)
    assert_close(d, want_d, rtol=1e-5, atol_scale=0, name="depth")
    assert_close(s, want_s, rtol=1e-4, atol_scale=1e-5, name="std")


def test_rays_and_samples(ops, enerf_fx):
    b = enerf_fx.batch(DEV)
    c = tiny_cfg(enerf_fx).enerf.cas_config
    H, W = b["src_inps"].shape[-2:]
    for lvl, inv in ((0, True), (1, False)):
        rs = c.render_scale[lvl]
        rays = ops.build_rays(b[f"rays_{lvl}"], g(enerf_fx, f"cap/depth_regression#{lvl}.0"),
                              g(enerf_fx, f"cap/depth_regression#{lvl}.1"), g(enerf_fx, f"cap/get_depth_values#{lvl}.1"),
                              int(H * rs), int(W * rs), inv)
        assert_close(rays, enerf_fx.t(f"cap/build_rays#{lvl}"), rtol=1e-5, atol_scale=1e-6, name=f"rays{lvl}")
        xyz, uvd, z = ops.sample_along_depth(g(enerf_fx, f"cap/build_rays#{lvl}"), c.num_samples[lvl], inv)
        assert_close(xyz, enerf_fx.t(f"cap/sample_along_depth#{lvl}.0"), rtol=1e-5, atol_scale=1e-6, name="xyz")
        assert_close(uvd, enerf_fx.t(f"cap/sample_along_depth#{lvl}.1"), rtol=1e-4, atol_scale=1e-5, name="uvd")
        assert_close(z, enerf_fx.t(f"cap/sample_along_depth#{lvl}.2"), rtol=1e-5, atol_scale=1e-6, name="z")


def test_lookups(ops, enerf_fx):
    b = enerf_fx.batch(DEV)
    c = tiny_cfg(enerf_fx).enerf.cas_config
    H, W = b["src_inps"].shape[-2:]
    feats = {0: g(enerf_fx, "cap/feature_net#0.0")[None], 2: g(enerf_fx, "cap/feature_net#0.2")[None]}
    for lvl in range(2):
        rs = c.render_scale[lvl]
        Hr, Wr = int(H * rs), int(W * rs)
        rgbs = ops.unpreprocess(b["src_inps"], Hr, Wr)
        assert_close(rgbs, enerf_fx.t(f"cap/unpreprocess#{lvl}"), rtol=1e-5, atol_scale=1e-6, name="unpre")
        uvd = g(enerf_fx, f"cap/sample_along_depth#{lvl}.1")
        uvd01 = torch.stack([uvd[..., 0] / (Wr - 1), uvd[..., 1] / (Hr - 1), uvd[..., 2]], -1).reshape(1, -1, 3).contiguous()
        vox = ops.vox_feat(uvd01, g(enerf_fx, f"cap/cost_reg_{lvl}#0.0"))
        assert_close(vox, enerf_fx.t(f"cap/get_vox_feat#{lvl}"), name=f"vox{lvl}")
        img = torch.cat([feats[c.render_im_feat_level[lvl]], g(enerf_fx, f"cap/unpreprocess#{lvl}")], 2).contiguous()
        feat = ops.img_feat(g(enerf_fx, f"cap/sample_along_depth#{lvl}.0"), img, b["src_exts"], b["src_ixts"], b["tar_ext"], rs)
        assert_close(feat, enerf_fx.t(f"cap/get_img_feat#{lvl}"), name=f"imgfeat{lvl}")


def _pack(ops, sd, prefix, feat_ch):
    tensors = []
    for name in ops.NERF_PARAM_ORDER:
        tensors += [sd[f"{prefix}{name}.weight"].contiguous(), sd[f"{prefix}{name}.bias"].contiguous()]
    return ops.nerf_pack_weights(tensors, feat_ch)


def test_mlp_and_composite(ops, enerf_fx):
    sd = enerf_fx.group("sd", DEV)
    c = tiny_cfg(enerf_fx).enerf.cas_config
    for lvl, feat_ch in ((1, 8), (0, 32)):
        blob = _pack(ops, sd, f"nerf_{lvl}.", feat_ch)
        raw = ops.nerf_mlp(g(enerf_fx, f"cap/get_vox_feat#{lvl}"), g(enerf_fx, f"cap/get_img_feat#{lvl}"), blob, feat_ch)
        assert_close(raw, enerf_fx.t(f"cap/nerf_{lvl}#0"), name=f"mlp{lvl}")
        Ns = c.num_samples[lvl]
        rgb, depth, weights = ops.composite(g(enerf_fx, f"cap/nerf_{lvl}#0").reshape(1, -1, Ns, 4),
                                            g(enerf_fx, f"cap/sample_along_depth#{lvl}.2"))
        assert_close(rgb, enerf_fx.t(f"cap/raw2outputs#{lvl}.rgb"), name="rgb")
        assert_close(depth, enerf_fx.t(f"cap/raw2outputs#{lvl}.depth"), name="depth")
        assert_close(weights, enerf_fx.t(f"cap/raw2outputs#{lvl}.weights"), name="weights")


def test_fused_render(ops, enerf_fx):
    sd = enerf_fx.group("sd", DEV)
    b = enerf_fx.batch(DEV)
    c = tiny_cfg(enerf_fx).enerf.cas_config
    H, W = b["src_inps"].shape[-2:]
    feats = {0: g(enerf_fx, "cap/feature_net#0.0")[None], 2: g(enerf_fx, "cap/feature_net#0.2")[None]}
    for lvl, feat_ch, inv in ((1, 8, False), (0, 32, True)):
        rs = c.render_scale[lvl]
        Hr, Wr = int(H * rs), int(W * rs)
        blob = _pack(ops, sd, f"nerf_{lvl}.", feat_ch)
        rgb_src = b["src_inps"] if rs == 1.0 else ops.unpreprocess(b["src_inps"], Hr, Wr)
        common = dict(feat_ch=feat_ch, Ns=c.num_samples[lvl], depth_inv=inv, Hr=Hr, Wr=Wr, render_scale=rs,
                      rgb_affine=(rs == 1.0))
        args = (b[f"rays_{lvl}"], g(enerf_fx, f"cap/depth_regression#{lvl}.0"), g(enerf_fx, f"cap/depth_regression#{lvl}.1"),
                g(enerf_fx, f"cap/get_depth_values#{lvl}.1"), g(enerf_fx, f"cap/cost_reg_{lvl}#0.0"),
                feats[c.render_im_feat_level[lvl]], rgb_src, b["src_exts"], b["src_ixts"], b["tar_ext"], blob)
        rgb, depth, weights = ops.render_rays(*args, mode=0, **common)
        assert_close(rgb, enerf_fx.t(f"cap/raw2outputs#{lvl}.rgb"), name=f"rgb{lvl}")
        assert_close(depth, enerf_fx.t(f"cap/raw2outputs#{lvl}.depth"), name=f"depth{lvl}")
        assert_close(weights, enerf_fx.t(f"cap/raw2outputs#{lvl}.weights"), name=f"weights{lvl}")
        raw, z, mask = ops.render_rays(*args, mode=1, **common)
        assert_close(raw.reshape(1, -1, 4), enerf_fx.t(f"cap/nerf_{lvl}#0"), name=f"raw{lvl}")
        assert_close(z, enerf_fx.t(f"cap/sample_along_depth#{lvl}.2"), rtol=1e-5, atol_scale=1e-6, name=f"z{lvl}")
        # ragged range: only rays [5, 77) are written
        N = b[f"rays_{lvl}"].shape[1]
        part = ops.render_rays(*args, mode=0, ray_range=(5, min(77, N)), **common)[0]
        assert_close(part[:, 5:min(77, N)], enerf_fx.t(f"cap/raw2outputs#{lvl}.rgb")[:, 5:min(77, N)], name="ragged")


def _network(fx, preset="enerf_eval"):
    from boostmvsnerfs_amd.config import set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    set_cfg(tiny_cfg(fx, preset))
    net = Network()
    net.load_state_dict(fx.group("sd"), strict=True)
    return net.to(DEV).eval()


@pytest.mark.parametrize("records", [True, False])
def test_network_forward_matches_reference(ops, enerf_fx, records):
    """The reference's output dict for the fixture batch, through both forms of the renderer's image lookups: 48-byte
    per-pixel records written by FeatureNet's fused last layer (the default at inference), and the planar maps."""
    from boostmvsnerfs_amd import convnet
    net = _network(enerf_fx)
    net.lookup_records = records
    seen = []
    real = ops.render_rays

    def spy(*a, **k):
        seen.append(k.get("im_packed") is not None)
        return real(*a, **k)
    ops.render_rays = spy
    try:
        with torch.no_grad():
            out = net(enerf_fx.batch(DEV))
    finally:
        ops.render_rays = real
    assert seen and seen[-1] == records                        # the form under test is the one the last level ran
    want = enerf_fx.group("out")
    assert set(out) == set(want)
    for k in want:
        assert_close(out[k], want[k], name=k)
    mse = float(((out["rgb_level1"].cpu() - want["rgb_level1"]) ** 2).mean())
    assert mse < 1e-8, f"PSNR delta too large (mse between renders {mse:.3e})"


def test_lookup_records_layout(ops, enerf_fx):
    """convnet.LookupRecords as FeatureNet writes them: record = [ch 0 2 4 6 | ch 1 3 5 7 | r b | g 0] of the planar
    8-channel map and the source image at that pixel."""
    net = _network(enerf_fx)
    x = enerf_fx.batch(DEV)["src_inps"][0]
    fn = net.feature_net
    with torch.no_grad():
        _, _, planar = fn(x)
        fn.pack_lookup = True
        try:
            _, _, rec = fn(x)
        finally:
            fn.pack_lookup = False
    t = rec.t
    assert t.shape == (x.shape[0], x.shape[2], x.shape[3], 12)
    p = planar.permute(0, 2, 3, 1)
    scale = float(p.abs().max())
    assert float((t[..., 0:4] - p[..., 0::2]).abs().max()) <= 1e-6 * scale
    assert float((t[..., 4:8] - p[..., 1::2]).abs().max()) <= 1e-6 * scale
    c = x.permute(0, 2, 3, 1)
    assert torch.equal(t[..., 8], c[..., 0]) and torch.equal(t[..., 9], c[..., 2]) and torch.equal(t[..., 10], c[..., 1])
    assert float(t[..., 11].abs().max()) == 0.0


def test_render_rays_and_batchify_keep_the_reference_signature(ops, enerf_fx):
    """lib/networks/enerf/network.py:24-56: render_rays(rays12, level=, batch=, im_feat=, feature_volume=, nerf_model=)
    and batchify_rays (chunks of cfg.enerf.chunk_size) on the reference's own captured inputs vs raw2outputs."""
    from boostmvsnerfs_amd.config import get_cfg
    net = _network(enerf_fx)
    c = get_cfg().enerf.cas_config
    b = enerf_fx.batch(DEV)
    g = lambda k: enerf_fx.t(k).to(DEV)
    feats = [g(f"cap/feature_net#0.{j}")[None] for j in range(3)]        # coarse -> fine
    for lvl in range(2):
        im_feat = feats[c.render_im_feat_level[lvl]]
        kw = dict(level=lvl, batch=b, im_feat=im_feat, feature_volume=g(f"cap/cost_reg_{lvl}#0.0"),
                  nerf_model=getattr(net, f"nerf_{lvl}"))
        rays12 = g(f"cap/build_rays#{lvl}")
        with torch.no_grad():
            out = net.render_rays(rays12, **kw)
            get_cfg().enerf.chunk_size = 100                             # ragged last chunk
            outb = net.batchify_rays(rays12, **kw)
        assert set(out) == {"rgb", "depth", "weights"}
        for k in out:
            assert_close(out[k], enerf_fx.t(f"cap/raw2outputs#{lvl}.{k}"), name=f"{k}{lvl}")
            assert torch.equal(out[k], outb[k]), k


def test_network_ray_sharding(ops, enerf_fx):
    net = _network(enerf_fx)
    from boostmvsnerfs_amd.config import get_cfg
    get_cfg().enerf.cas_config.render_if = [False, True]     # eval config: only level 1 is rendered
    b = enerf_fx.batch(DEV)
    with torch.no_grad():
        full = net(b)["rgb_level1"]
        N = b["rays_1"].shape[1]
        net.ray_range = (N // 2, N)
        half = net(b)["rgb_level1"]
    assert half.shape[1] == N - N // 2
    assert torch.equal(half, full[:, N // 2:])


def test_errors_are_loud(ops):
    x = torch.zeros(1, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.depth_values_uniform(x, 4, 2, 2, True)
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.nerf_pack_weights([torch.zeros(4, device=DEV)] * 16, 5)


def test_sweep_extreme_coordinates(ops):
    """Planes at ~0 depth send the warp to 1e6+ pixels: must give exact zeros, no NaN/overflow."""
    torch.manual_seed(0)
    feats = torch.randn(1, 3, 8, 16, 20, device=DEV)
    proj = torch.eye(4, device=DEV)[:3][None, None].repeat(1, 3, 1, 1).contiguous()
    proj[..., 3] = torch.tensor([5.0, -3.0, -0.5], device=DEV)   # z < 0 -> clamp 1e-6
    dv = torch.full((1, 4, 8, 10), 1e-9, device=DEV)
    var = ops.sweep_variance(feats, proj, dv, algo=1)
    assert torch.isfinite(var).all() and float(var.abs().max()) == 0.0
    # channel-last kernels (16 channels): the windowed kernel sees an EMPTY tap box for every view
    feats16 = torch.randn(1, 3, 16, 16, 20, device=DEV)
    for algo in (4, 41) + QUAD_ALGOS + QUAD_PU_ALGOS:
        var = ops.sweep_variance(feats16, proj, dv, algo=algo)
        assert torch.isfinite(var).all() and float(var.abs().max()) == 0.0, algo


QUAD_ALGOS = tuple(range(500, 517))     # csrc/sweep_quad.hip: quad-planar features, union windows; tuning variants
QUAD_PU_ALGOS = tuple(range(600, 617))  # ... told that every plane of the hypotheses is constant (cascade level 0)


def _quad(ops, feats, channels_last=False):
    return ops.QuadFeats(ops.to_quad_planar(feats, channels_last=channels_last))


@pytest.mark.parametrize("level", [0, 1])
def test_sweep_kernels_agree_at_scale(ops, level):
    """All sweep kernels (reference-layout gather, windowed channel-last, quad-planar union windows: every tuning variant)
    against the CPU oracle on BASELINE config-1 shapes (256x320), including a wide-baseline view whose epipolar slide
    leaves the LDS windows (global fallback path), LDS budgets below the windows, plane-uniform hypotheses in both
    forms and views picked by index."""
    from boostmvsnerfs_amd.synthetic import make_batch
    from oracle import enerf as O
    H, W = 256, 320
    b = make_batch(H, W, seed=5)
    b["src_exts"][0, 2, 0, 3] += 2.5           # view 2: large baseline -> long epipolar segments
    cfgl = {0: dict(C=32, fs=0.25, vs=0.125, D=32), 1: dict(C=16, fs=0.5, vs=0.5, D=8)}[level]
    Hs, Ws, h, w = int(H * cfgl["fs"]), int(W * cfgl["fs"]), int(H * cfgl["vs"]), int(W * cfgl["vs"])
    torch.manual_seed(level)
    feats = torch.randn(1, 3, cfgl["C"], Hs, Ws)
    P = O.proj_mats(b["src_exts"], b["src_ixts"], b["tar_ext"], b["tar_ixt"], cfgl["fs"], cfgl["vs"])
    if level == 0:
        dv, _ = O.depth_hypotheses_uniform(torch.tensor([[0.4, 8.0]]), cfgl["D"], h, w, True)
    else:
        dv = (3.0 + 2.0 * torch.rand(1, 1, h, w) + torch.linspace(-1.5, 1.5, cfgl["D"]).view(1, -1, 1, 1)).contiguous()
    want = O.variance_volume(feats, P, dv)
    fd, Pd, dvd = feats.to(DEV), P.to(DEV), dv.to(DEV)
    for algo in (1, 4, 0) + tuple(range(40, 60)) + QUAD_ALGOS + (QUAD_PU_ALGOS if level == 0 else ()):
        got = ops.sweep_variance(fd, Pd, dvd, algo=algo)
        assert_close(got, want, name=f"level {level} algo {algo}")
    # windowed kernel with an LDS budget far below the tap boxes: clipped windows + the global fallback per wave
    from boostmvsnerfs_amd import _lib
    _lib.set_tuning("BMV_SWEEP_WIN_CAP", 48)
    try:
        for algo in (4, 41, 49):
            assert_close(ops.sweep_variance(fd, Pd, dvd, algo=algo), want, name=f"level {level} algo {algo} cap 48")
    finally:
        _lib.set_tuning("BMV_SWEEP_WIN_CAP", None)
    # quad-planar kernel with LDS budgets below the union windows: 1 KB = no view fits (every view gathered from global
    # memory), 6 / 12 KB = some views staged, the rest gathered, in one workgroup
    q = _quad(ops, fd)
    for variant in (0, 12, 3, 14, 17, 18, 19, 20):     # (17-20: a workgroup walks 2 / all plane groups of its tile)
        for budget in (1, 6, 12):
            got = ops.sweep_variance_quad(q, Pd, dvd, variant=variant, flags=budget << 16)
            assert_close(got, want, name=f"level {level} quad variant {variant} budget {budget} KB")
    if level == 0:     # hypotheses handed over as (B, D): one per plane (dv_plane_uniform 1)
        got = ops.sweep_variance_quad(q, Pd, dvd[:, :, 0, 0].contiguous(), hw=(h, w))
        assert_close(got, want, name="level 0 quad, (B, D) hypotheses")
    # the variance as quad records (what the regulariser's first layer stages with 16-byte loads): the same values, bit
    # for bit, in another layout -- default variants, small LDS budgets (gather fallback) included; a tuning variant
    # without that output falls back to the planar tensor
    for variant, budget in ((-1, 0), (0, 0), (12, 0), (12, 6), (0, 1)):
        planar = ops.sweep_variance_quad(q, Pd, dvd, variant=variant, flags=budget << 16)
        qv = ops.sweep_variance_quad(q, Pd, dvd, variant=variant, flags=budget << 16, quad_out=True)
        assert isinstance(qv, ops.QuadVolume) and qv.shape == planar.shape
        assert torch.equal(qv.to_planar(), planar), f"level {level} variant {variant}: quad records differ from the planar volume"
    assert torch.is_tensor(ops.sweep_variance_quad(q, Pd, dvd, variant=3, quad_out=True))
    # views picked by index from a larger set (the K-volume networks)
    extra = torch.randn(1, 2, cfgl["C"], Hs, Ws, device=DEV)
    allv = torch.cat([extra[:, :1], fd[:, 2:3], fd[:, 0:1], extra[:, 1:], fd[:, 1:2]], 1)      # views 2, 4, 1 are ours
    ids = torch.tensor([[2, 4, 1]], device=DEV, dtype=torch.int32)
    got = ops.sweep_variance_views(_quad(ops, allv), ids, Pd, dvd, plane_uniform=level == 0)
    assert_close(got, want, name=f"level {level} quad, views by index")


@pytest.mark.parametrize("shape", [(1, 2, 16, 37, 53, 5, 19, 45), (2, 4, 32, 40, 24, 7, 21, 13), (1, 3, 16, 9, 7, 3, 33, 70),
                                   (1, 3, 16, 150, 20, 4, 150, 14)])    # one tile column, many tile rows per XCD band
def test_sweep_windowed_ragged_shapes(ops, shape):
    """Windowed kernel on sizes that are not multiples of its tiles (partial tiles in x, y and planes), 2 and 4
    views, batch 2, a source smaller than one window piece, vs the CPU oracle."""
    from oracle import enerf as O
    B, S, C, Hs, Ws, D, h, w = shape
    torch.manual_seed(sum(shape))
    feats = torch.randn(B, S, C, Hs, Ws)
    P = torch.zeros(B, S, 3, 4)
    for b in range(B):
        for s in range(S):
            sx, sy = Ws / w, Hs / h
            P[b, s] = torch.tensor([[sx, 0.05 * s, 0.3 * s, 4.0 * (s - 1)], [-0.04 * s, sy, 0.2 * b, 3.0 * (1 - s)],
                                    [0.0, 0.0, 1.0, 0.05 * s]])
    dv = (2.0 + torch.rand(B, 1, h, w) + torch.linspace(0.0, 3.0, D).view(1, -1, 1, 1)).contiguous()
    want = O.variance_volume(feats, P, dv)
    for algo in (4, 40, 42, 46, 49, 51, 57, 59) + QUAD_ALGOS:
        got = ops.sweep_variance(feats.to(DEV), P.to(DEV), dv.to(DEV), algo=algo)
        assert_close(got, want, name=f"shape {shape} algo {algo}")


@pytest.mark.parametrize("C,S", [(4, 2), (8, 3), (12, 4), (64, 2)])
def test_sweep_quad_channel_counts(ops, C, S):
    """The quad-planar kernel takes any C % 4 == 0 up to 64 and 2..4 views (the windowed kernel: 16 / 32 channels only)."""
    from oracle import enerf as O
    B, Hs, Ws, D, h, w = 1, 40, 56, 6, 20, 28
    torch.manual_seed(C * S)
    feats = torch.randn(B, S, C, Hs, Ws)
    P = torch.zeros(B, S, 3, 4)
    for s_ in range(S):
        P[0, s_] = torch.tensor([[2.0, 0.03 * s_, 0.2 * s_, 3.0 * (s_ - 1)], [-0.02 * s_, 2.0, 0.1, 2.0 * (1 - s_)], [0.0, 0.0, 1.0, 0.04 * s_]])
    dv = (2.0 + torch.rand(B, 1, h, w) + torch.linspace(0.0, 2.0, D).view(1, -1, 1, 1)).contiguous()
    want = O.variance_volume(feats, P, dv)
    for variant in (-1, 0, 1, 3, 12, 14):
        got = ops.sweep_variance_quad(_quad(ops, feats.to(DEV)), P.to(DEV), dv.to(DEV), variant=variant)
        assert_close(got, want, name=f"C {C} S {S} variant {variant}")
        if variant in (-1, 0, 12):     # ... and as quad records (ragged tiles, every view count): the same bits
            qv = ops.sweep_variance_quad(_quad(ops, feats.to(DEV)), P.to(DEV), dv.to(DEV), variant=variant, quad_out=True)
            assert isinstance(qv, ops.QuadVolume) and torch.equal(qv.to_planar(), got)
    with pytest.raises(RuntimeError, match="not covered"):
        ops.sweep_variance_quad(torch.zeros(1, 5, 1, 8, 8, 4, device=DEV), torch.zeros(1, 5, 3, 4, device=DEV), dv.to(DEV))
    # BMV_SWEEP_QUAD_DBL=1 (round 6, opt-in): two window sets per workgroup, the next quad's fill under this quad's blend,
    # hand-written LDS reads -- at this size the sets fit twice, so the path runs for every C > 4.  Equal to fp32 rounding,
    # not bit for bit: the compiler contracts the blend's multiply-adds differently in the two copies of the quad loop
    # (measured: <= 3e-7 of the volume's scale on ~1-2 % of the entries, every one of them in the second plane of a lane's
    # pair, identical from run to run); against the oracle both forms pass the same bar
    from boostmvsnerfs_amd import _lib
    ref = {v: ops.sweep_variance_quad(_quad(ops, feats.to(DEV)), P.to(DEV), dv.to(DEV), variant=v, quad_out=True).to_planar() for v in (-1, 0, 12)}
    _lib.set_tuning("BMV_SWEEP_QUAD_DBL", 1)
    try:
        for v, want_v in ref.items():
            got = ops.sweep_variance_quad(_quad(ops, feats.to(DEV)), P.to(DEV), dv.to(DEV), variant=v, quad_out=True).to_planar()
            scale = float(want_v.abs().max())
            assert float((got - want_v).abs().max()) <= 1e-6 * scale, f"C {C} S {S} variant {v}: two window sets"
            assert_close(got, want, name=f"C {C} S {S} variant {v}, two window sets")
            got2 = ops.sweep_variance_quad(_quad(ops, feats.to(DEV)), P.to(DEV), dv.to(DEV), variant=v)
            assert torch.equal(got2, got), f"C {C} S {S} variant {v}: two window sets, planar output = quad records"
    finally:
        _lib.set_tuning("BMV_SWEEP_QUAD_DBL", None)


def test_tuning_switches_are_library_state():
    """include/bmv.h bmv_tuning_*: explicit, listable, changeable at any time; unknown names are errors."""
    from boostmvsnerfs_amd import _lib
    names = _lib.tuning_names()
    assert "BMV_RENDER_PC" in names and "BMV_SWEEP_WIN_CAP" in names and len(names) >= 10
    assert _lib.get_tuning("BMV_CONV0_R") is None
    _lib.set_tuning("BMV_CONV0_R", 8)
    assert _lib.get_tuning("BMV_CONV0_R") == 8
    _lib.set_tuning("BMV_CONV0_R", None)
    assert _lib.get_tuning("BMV_CONV0_R") is None
    with pytest.raises(RuntimeError, match="unknown switch"):
        _lib.set_tuning("BMV_NO_SUCH_SWITCH", 1)


def test_make_rays_matches_reference_rays():
    """bmv_make_rays vs rays the REFERENCE's build_rays (lib/datasets/enerf_utils.py:25-71) produced
    (tests/golden/rays_tiny.npz): two cameras (one with the principal point off the pixel grid), both render scales."""
    import os
    import numpy as np
    from boostmvsnerfs_amd import ops
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rays_tiny.npz"))
    for c in range(2):
        H, W = (int(v) for v in fx[f"in/hw_{c}"])
        ext = torch.from_numpy(fx[f"in/tar_ext_{c}"])[None].to(DEV)
        ixt = torch.from_numpy(fx[f"in/tar_ixt_{c}"])[None].to(DEV)
        for level in range(2):
            want = torch.from_numpy(fx[f"out/rays_{c}_{level}"])[None]
            got = ops.make_rays(ext, ixt, H, W, float(fx[f"extra/scale_{level}"])).cpu()
            assert got.shape == want.shape
            assert torch.equal(got[..., 6:], want[..., 6:])
            # float64 inside vs the reference's float32 inverses: a few float32 ulp of the largest component
            assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_forward_builds_missing_rays_on_the_device(enerf_fx):
    """A batch without rays_i (Network.ensure_rays -> bmv_make_rays) renders what the batch with the loader's rays does."""
    from boostmvsnerfs_amd.config import set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    set_cfg(tiny_cfg(enerf_fx))
    net = Network()
    net.load_state_dict(enerf_fx.group("sd"), strict=True)
    net = net.to(DEV).eval()
    b = enerf_fx.batch(DEV)
    with torch.no_grad():
        want = net(b)
        b2 = {k: v for k, v in enerf_fx.batch(DEV).items() if not k.startswith("rays_")}
        got = net(b2)
    assert "rays_1" in b2 and torch.equal(b2["rays_1"][..., 6:], b["rays_1"][..., 6:])
    for k in want:
        assert_close(got[k], want[k], rtol=1e-4, atol_scale=1e-5, name=k)


def test_get_ndc_coords():
    from boostmvsnerfs_amd.networks.enerf import utils as U
    from boostmvsnerfs_amd.synthetic import make_batch
    from oracle import enerf as O
    b = make_batch(48, 64, n_views=3, seed=2, B=2)
    torch.manual_seed(0)
    xyz = torch.randn(2, 37, 4, 3) * 0.5 + torch.tensor([0.0, 0.0, 4.0])
    inv = torch.tensor([[63.0, 47.0]]).expand(2, -1)
    want = O.ndc_coords(xyz, b["src_exts"][:, 1], b["src_ixts"][:, 1], inv)
    got = U.get_ndc_coords(xyz.to(DEV), b["src_exts"][:, 1].to(DEV), b["src_ixts"][:, 1].to(DEV), inv.to(DEV))
    assert_close(got, want, rtol=1e-5, atol_scale=1e-6, name="ndc")
    # and mask_viewport is the viewport test on exactly these coordinates
    m = U.mask_viewport(xyz.to(DEV), b["src_exts"][:, 1:2].to(DEV), b["src_ixts"][:, 1:2].to(DEV), inv.to(DEV)).cpu()
    vis = ((want[..., 0] >= 0) & (want[..., 0] <= 1) & (want[..., 1] >= 0) & (want[..., 1] <= 1) & (want[..., 2] > 0))
    assert float((m.reshape(vis.shape) - vis.float()).abs().mean()) < 1e-3


def test_make_rays_matches_the_dataset_ray_builder():
    """bmv_make_rays vs the numpy restatement of `build_rays` (lib/datasets/enerf_utils.py:25-31, 62-71) the synthetic
    batches are built with: both render scales, two different target cameras in one batch."""
    import numpy as np
    from boostmvsnerfs_amd import ops
    from boostmvsnerfs_amd.synthetic import make_batch, make_rays
    H, W = 96, 160
    b = make_batch(H, W, n_views=3, seed=3, B=2)
    ext, ixt = b["tar_ext"].numpy().astype(np.float64), b["tar_ixt"].numpy().astype(np.float64)
    for scale in (1.0, 0.25):
        got = ops.make_rays(b["tar_ext"].to(DEV), b["tar_ixt"].to(DEV), H, W, scale).cpu()
        want = torch.from_numpy(np.stack([make_rays(ext[i], ixt[i], H, W, scale) for i in range(2)]))
        assert got.shape == want.shape == (2, int(H * scale) * int(W * scale), 8)
        assert torch.equal(got[..., 6:], want[..., 6:])                       # pixel coordinates
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
        # and they are what the batch carries
        key = "rays_1" if scale == 1.0 else "rays_0"
        assert float((got - b[key]).abs().max()) <= 1e-6 * float(want.abs().max())


def test_path_selection_follows_what_can_receive_a_gradient(ops, enerf_fx):
    """Grad mode alone does not switch to the op-by-op path: a frozen network called without torch.no_grad() keeps the
    fused kernels (so chunking and ray sharding keep working); the differentiable path honours ray_range too."""
    from boostmvsnerfs_amd.config import get_cfg
    net = _network(enerf_fx)
    get_cfg().enerf.cas_config.render_if = [False, True]
    b = enerf_fx.batch(DEV)
    with torch.no_grad():
        want = net(b)["rgb_level1"]
    N = b["rays_1"].shape[1]
    for p in net.parameters():
        p.requires_grad_(False)
    assert not net.wants_grad()
    net.ray_range = (N // 3, N)
    frozen = net(enerf_fx.batch(DEV))["rgb_level1"]              # grad mode on, nothing to differentiate: fused path
    assert not frozen.requires_grad and frozen.shape[1] == N - N // 3
    assert_close(frozen, want[:, N // 3:], rtol=1e-5, atol_scale=1e-6, name="frozen")
    for p in net.parameters():
        p.requires_grad_(True)
    assert net.wants_grad()
    part = net(enerf_fx.batch(DEV))["rgb_level1"]                # differentiable path, same ray slice
    assert part.requires_grad and part.shape[1] == N - N // 3
    assert_close(part, want[:, N // 3:], name="train path, ray_range")


def test_render_pc_lost_wakeup_surfaces_as_an_error(enerf_fx):
    """The producer / consumer renderer's waits are bounded: a wave that never sees its mailbox flag gives up (its pixels
    stay unwritten) instead of hanging the GPU -- and that MUST NOT stay silent: the waves count, bmv_render_pc_check
    reports the count through bmv_last_error.  bmv_debug_render_pc_inject withholds one wake-up in workgroup 0."""
    from boostmvsnerfs_amd import _lib
    lib = _lib.load()
    net = _network(enerf_fx)
    b = enerf_fx.batch(DEV)
    with torch.no_grad():
        want = {k: v.clone() for k, v in net._forward_checked(dict(b)).items()}
    torch.cuda.synchronize()
    assert lib.bmv_render_pc_check(1) == 0                      # a clean frame: no wave gave up
    assert lib.bmv_debug_render_pc_inject(1) == 0
    try:
        with torch.no_grad():
            net._forward_checked(dict(b))
        torch.cuda.synchronize()
        rc = lib.bmv_render_pc_check(1)
        msg = lib.bmv_last_error().decode()
        assert rc != 0 and "lost wake-up" in msg and "gave up" in msg, (rc, msg)
        with pytest.raises(RuntimeError, match="lost wake-up"):
            lib.bmv_debug_render_pc_inject(1)
            with torch.no_grad():
                net._forward_checked(dict(b))
            _lib.check(lib.bmv_render_pc_check(1), "render_pc_check")
    finally:
        assert lib.bmv_debug_render_pc_inject(0) == 0
    with torch.no_grad():
        got = net._forward_checked(dict(b))
    torch.cuda.synchronize()
    assert lib.bmv_render_pc_check(1) == 0
    for k in want:
        assert torch.equal(got[k], want[k]), k


@pytest.mark.parametrize("S", [2, 4])
def test_enerf_with_2_and_4_source_views(enerf_fx, S):
    """ENeRF with 2 and 4 source views (the reference trains with train_input_views [2, 3, 4],
    configs/exps/pretrain/enerf/dtu_pretrain.yaml:22-23, 75-76): Network.forward (eval) against the reference's own output
    dict, and every parameter gradient of the fine-tune loss against the reference's (tests/golden/enerf_tiny_views{S}.npz).
    Everything runs on the HIP kernels' S = 2 / 4 instantiations (sweeps, fused renderer, MLP forward / backward): the
    profiler must not see a single GEMM launch (round 4 ran the MLP of S != 3 as torch ops on rocBLAS)."""
    from conftest import check_param_grads, load_fixture
    from boostmvsnerfs_amd.config import set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.train import NetworkWrapper
    vfx = load_fixture(f"enerf_tiny_views{S}")
    cfg = tiny_cfg(enerf_fx, "enerf_pretrain")
    cfg.enerf.cas_config.render_if = [True, True]
    set_cfg(cfg)
    net = Network()
    net.load_state_dict(enerf_fx.group("sd"), strict=True)
    net = net.to(DEV).eval()
    bg = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in vfx.batch().items()}
    assert bg["src_inps"].shape[1] == S
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        with torch.no_grad():
            out = net({k: (v.clone() if torch.is_tensor(v) else v) for k, v in bg.items()})
        net.zero_grad()
        _, loss, _, _ = NetworkWrapper(net)(bg)
        loss.backward()
        torch.cuda.synchronize()
    for k, v in vfx.group("out").items():
        assert_close(out[k], v, name=f"S={S} {k}")
    ref = {k[5:]: torch.from_numpy(v) for k, v in vfx.raw.items() if k.startswith("grad/")}
    check_param_grads(net, ref, float(loss), float(vfx.raw["extra/loss"]))
    kernels = {e.key for e in prof.key_averages() if getattr(e, "device_type", None) is not None and
               str(e.device_type).endswith("CUDA")}
    assert any("render" in k for k in kernels) and any("nerf_mlp_bwd" in k for k in kernels), sorted(kernels)[:40]
    gemms = [k for k in kernels if any(t in k.lower() for t in ("gemm", "cijk", "rocblas", "hipblas"))]
    assert not gemms, f"S={S}: GEMM launches on the path: {gemms}"


@pytest.mark.parametrize("n_views,num_samples,render_if", [(2, [8, 1], [False, True]), (2, [8, 4], [False, True]),
                                                          (2, [4, 8], [True, True]), (4, [8, 1], [False, True]),
                                                          (4, [8, 4], [False, True]), (4, [2, 8], [True, True])])
@pytest.mark.parametrize("records", [True, False])
def test_two_and_four_views_at_every_sample_count(n_views, num_samples, render_if, records):
    """with S = 2 / 4 source views only the shipped sample counts (2 at level 1, 8 at level 0) had fused
    renderer kernels; users vary `test_input_views` and `num_samples` independently (lib/networks/enerf/network.py:24-43
    works for any combination).  Whole frames of `Network.forward` at S in {2, 4} x Ns in {1, 4, 8} (level 1: feat_ch 8)
    and Ns in {2, 4} (level 0: feat_ch 32, inverse depth), through the lookup-record (producer / consumer) and the
    planar-lookup renderers, against the oracle's frame on the same weights and batch."""
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    from oracle import enerf as O
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [16, 8]
    cfg.enerf.cas_config.num_samples = list(num_samples)
    cfg.enerf.cas_config.render_if = list(render_if)
    set_cfg(cfg)
    torch.manual_seed(0)
    net = Network().eval()
    with torch.no_grad():
        for i in range(2):
            getattr(net, f"cost_reg_{i}").depth_conv[0].weight.mul_(40.0)
    batch = make_batch(64, 96, n_views=n_views, seed=5)
    with torch.no_grad():
        want = O.enerf_forward({k: v.clone() for k, v in net.state_dict().items()}, clone_batch(batch), cfg)
        net = net.to(DEV)
        net.lookup_records = records
        got = net._forward_checked(clone_batch(batch, DEV))
    levels = [i for i in range(2) if render_if[i]]
    assert levels and all(f"rgb_level{i}" in got for i in levels)
    for i in levels:
        for k in (f"rgb_level{i}", f"depth_level{i}"):
            assert_close(got[k], want[k], name=f"S={n_views} Ns={num_samples} records={records} {k}")


@pytest.mark.parametrize("n_views", [3, 2, 4])
def test_renderer_split_bf16_experiment_is_fp32_equivalent(n_views):
    """bmv_tuning BMV_RENDER_SPLIT (default 1 since the end of round 5; 0 = every chain on fp32 MFMAs): the MLP's two-tile
    chains -- 160 of its 206 matrix instructions per tile -- on the bf16 matrix pipe with BOTH operands split into three
    bf16 pieces (the fp32 values exactly; the product terms dropped are at most 2^-23 of a product).  The frame it renders
    must agree with the all-fp32-MFMA frame to fp32 rounding (not bit for bit: the summation order differs), far inside
    the 1e-3 bar of the parity tests."""
    from boostmvsnerfs_amd import _lib
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [32, 8]
    set_cfg(cfg)
    torch.manual_seed(3)
    net = Network().eval().to(DEV)
    batch = clone_batch(make_batch(128, 160, n_views=n_views, seed=3), DEV)
    was = _lib.get_tuning("BMV_RENDER_SPLIT")      # (the suite may be running with the experiment switched on)
    with torch.no_grad():
        try:
            _lib.set_tuning("BMV_RENDER_SPLIT", 0)
            ref = net._forward_checked(dict(batch))
            _lib.set_tuning("BMV_RENDER_SPLIT", 1)
            got = net._forward_checked(dict(batch))
        finally:
            _lib.set_tuning("BMV_RENDER_SPLIT", was)
    assert not torch.equal(got["rgb_level1"], ref["rgb_level1"]), "the split path did not run"
    for k in ("rgb_level1", "depth_level1", "weights_level1"):
        d = float((got[k] - ref[k]).abs().max())
        scale = float(ref[k].abs().max())
        assert d <= 2e-6 * scale, f"{k}: {d:.3e} against scale {scale:.3e}"


@pytest.mark.parametrize("records", [True, False])
def test_regulariser_first_layers_and_heads_on_the_bf16_pipe_are_fp32_equivalent(enerf_fx, records):
    """BMV_CONV_C4S (csrc/conv_c4s.hip): the regularisers' first layers and heads -- the four matrix-bound layers of a
    frame -- as bf16 MFMAs on three-piece fp32 operands.  The frame must (i) match the reference's output dict at the
    project tolerance in both forms, (ii) agree with the fp32-block form to fp32 rounding (5e-6 of an output's scale), and
    (iii) not be bit-equal to it (the split kernels ran: four launches per frame are counted)."""
    from boostmvsnerfs_amd import convnet
    net = _network(enerf_fx)
    net.lookup_records = records
    want = enerf_fx.group("out")
    frames, calls = {}, []
    real = convnet.conv_c4s_fwd

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    convnet.conv_c4s_fwd = spy
    try:
        for on in (False, True):
            for i in range(2):
                getattr(net, f"cost_reg_{i}").conv_c4s = on
            with torch.no_grad():
                frames[on] = net._forward_checked(enerf_fx.batch(DEV))
            for k in want:
                assert_close(frames[on][k], want[k], name=f"{k}, conv_c4s={on}")
    finally:
        convnet.conv_c4s_fwd = real
    assert len(calls) == 4, calls          # conv0 + heads of both regularisers, in the split form's frame only
    differs = False
    for k in want:
        a, b_ = frames[False][k], frames[True][k]
        differs |= not torch.equal(a, b_)
        d, scale = float((a - b_).abs().max()), float(a.abs().max())
        print(f"[conv_c4s frame, records={records}] {k}: max |d| {d:.3e} (scale {scale:.3e})")
        # 5e-6 of the output's scale (measured: up to 2.1e-6 on rgb_level1 -- the regularisers' rounding differences travel
        # through the depth distribution, the sample placement and the MLP -- and 0.6e-6 on the depth maps)
        assert d <= 5e-6 * scale, f"{k}: {d:.3e} against scale {scale:.3e}"
    assert differs, "the split path did not run"


def test_split_mlp_is_as_accurate_as_the_fp32_mlp_against_float64():
    """The experiment's arithmetic claim, as a test: on the same fp32 inputs and weights the MLP with its two-tile chains
    on the bf16 pipe (three-piece operands) is no farther from a float64 evaluation of the network
    (oracle/enerf.py nerf_mlp, lib/networks/enerf/nerf.py:29-43, 74-89) than the fp32-MFMA form is."""
    from boostmvsnerfs_amd import _lib, ops
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from oracle import enerf as O
    set_cfg(make_cfg("enerf_eval"))
    from boostmvsnerfs_amd.networks.enerf.nerf import NeRF
    torch.manual_seed(11)
    P = 1 << 14
    net = NeRF(feat_ch=8 + 3)
    with torch.no_grad():
        for p_ in net.parameters():
            if p_.dim() == 1:
                p_.normal_(0, 0.1)
    sd = {("nerf." + k): v.detach().double() for k, v in net.state_dict().items()}
    vox = torch.randn(1, P, 8)
    img = torch.cat([torch.randn(1, P, 3, 8), torch.rand(1, P, 3, 3), torch.randn(1, P, 3, 4) * 0.5], -1)
    want = O.nerf_mlp(sd, "nerf.", vox.double(), img.double())
    netd = net.to(DEV).eval()
    err = {}
    with torch.no_grad():
        was = _lib.get_tuning("BMV_RENDER_SPLIT")
        for split in (0, 1):
            _lib.set_tuning("BMV_RENDER_SPLIT", split)
            try:
                got = ops.nerf_mlp(vox.to(DEV), img.to(DEV), netd.packed_weights(), 8).cpu().double()
            finally:
                _lib.set_tuning("BMV_RENDER_SPLIT", was)
            err[split] = (got - want).abs()
    assert float((err[1] - err[0]).abs().max()) > 0, "the split path did not run"
    for name, sl in (("rgb", slice(0, 3)), ("sigma", slice(3, 4))):
        e0, e1 = err[0][..., sl], err[1][..., sl]
        assert float(e1.mean()) <= 1.5 * float(e0.mean()) + 1e-9, (name, float(e1.mean()), float(e0.mean()))
        assert float(e1.max()) <= 2.0 * float(e0.max()) + 1e-8, (name, float(e1.max()), float(e0.max()))
