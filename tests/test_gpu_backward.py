"""Backward kernels vs torch.autograd on the CPU oracle (fine-tuning path, BASELINE config 5)."""
import pytest
import torch

from conftest import assert_close, tiny_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"
GTOL = dict(rtol=2e-3, atol_scale=2e-3)    # gradients: sums of many products, scatter-add order not fixed


def leaf(t):
    return t.detach().clone().requires_grad_(True)


def test_composite_and_blend_bwd():
    from boostmvsnerfs_amd import autograd as A
    from oracle import enerf as O
    torch.manual_seed(0)
    for Ns in (2, 8, 32):
        raw = torch.rand(1, 301, Ns, 4) * torch.tensor([1, 1, 1, 3.0])
        z = torch.rand(1, 301, Ns) + 2
        g_rgb, g_depth = torch.randn(1, 301, 3), torch.randn(1, 301)
        r = leaf(raw)
        out = O.composite(r, z)
        (out["rgb"] * g_rgb).sum().add((out["depth"] * g_depth).sum()).backward()
        rg = leaf(raw.to(DEV))
        rgb, depth, _ = A.Composite.apply(rg, z.to(DEV))
        ((rgb * g_rgb.to(DEV)).sum() + (depth * g_depth.to(DEV)).sum()).backward()
        assert_close(rg.grad, r.grad, name=f"d_raw Ns={Ns}", **GTOL)
    K, N, Ns = 3, 257, 4
    raws = torch.rand(2, K, N, Ns, 4) * torch.tensor([1, 1, 1, 2.0])
    masks = O.normalise_masks(torch.randint(0, 4, (2, K, N, Ns)).float() / 3)
    z = torch.rand(2, K, N, Ns) + 2
    g = torch.randn(2, N, 3)
    r = leaf(raws)
    (O.blend(r, masks, z)["rgb"] * g).sum().backward()
    rg = leaf(raws.to(DEV))
    (A.Blend.apply(rg, masks.to(DEV), z.to(DEV))[0] * g.to(DEV)).sum().backward()
    assert_close(rg.grad, r.grad, name="d_raws", **GTOL)


def rays_are_full_frame(rays, Hr, Wr):
    return rays.shape[1] == Hr * Wr


def test_lookup_and_sampler_bwd(enerf_fx, scatter_mode):
    from boostmvsnerfs_amd import autograd as A
    from oracle import enerf as O
    b = enerf_fx.batch()
    c = tiny_cfg(enerf_fx).enerf.cas_config
    H, W = b["src_inps"].shape[-2:]
    torch.manual_seed(1)
    for lvl, inv in ((1, False), (0, True)):
        rs = c.render_scale[lvl]
        Hr, Wr = int(H * rs), int(W * rs)
        depth, std = enerf_fx.t(f"cap/depth_regression#{lvl}.0"), enerf_fx.t(f"cap/depth_regression#{lvl}.1") * 0.3
        nf = enerf_fx.t(f"cap/get_depth_values#{lvl}.1")
        vol = enerf_fx.t(f"cap/cost_reg_{lvl}#0.0")
        img = torch.cat([enerf_fx.t(f"cap/feature_net#0.{0 if lvl == 0 else 2}")[None], enerf_fx.t(f"cap/unpreprocess#{lvl}")], 2)
        Ns = c.num_samples[lvl]

        def run(dev, mods, hinted=False):
            d, s, v, im = (leaf(t.to(dev)) for t in (depth, std, vol, img))
            bb = {k: (t.to(dev) if torch.is_tensor(t) else t) for k, t in b.items()}
            if mods is O:
                rays = O.rays_with_bounds(bb[f"rays_{lvl}"], d, s, nf.to(dev), rs / c.volume_scale[lvl], inv)
                xyz, uvd, z = O.sample_points(rays, Ns, inv)
                uvd01 = torch.stack([uvd[..., 0] / (Wr - 1), uvd[..., 1] / (Hr - 1), uvd[..., 2]], -1).reshape(1, -1, 3)
                vox = O.vox_lookup(uvd01, v)
                feat = O.img_lookup(xyz, im, bb["src_exts"], bb["src_ixts"], bb["tar_ext"], rs)
            else:
                rays = mods.BuildRays.apply(bb[f"rays_{lvl}"], d, s, nf.to(dev), Hr, Wr, inv)
                xyz, uvd, z = mods.SampleAlongDepth.apply(rays, Ns, inv)
                uvd01 = torch.stack([uvd[..., 0] / (Wr - 1), uvd[..., 1] / (Hr - 1), uvd[..., 2]], -1).reshape(1, -1, 3)
                vox = mods.VoxFeat.apply(uvd01, v, *((Wr, Ns) if hinted else ()))
                # hinted: tell the backward that only the feature channels need d_img and that the rays are a
                # full Hr x Wr frame (2-D tiles + LDS pre-reduction instead of 256 consecutive samples)
                hints = (im.shape[2] - 3, Wr) if hinted else ()
                feat = mods.ImgFeat.apply(xyz, im, bb["src_exts"], bb["src_ixts"], bb["tar_ext"], rs, *hints)
            torch.manual_seed(7)
            gv, gf = torch.randn(vox.shape), torch.randn(feat.shape)
            ((vox * gv.to(dev)).sum() + (feat * gf.to(dev)).sum()).backward()
            return d.grad, s.grad, v.grad, im.grad

        want = run("cpu", O)
        got = run(DEV, A)
        for name, g, w_ in zip(("d_depth", "d_std", "d_volume", "d_img"), got, want):
            assert_close(g, w_, name=f"{name} level {lvl}", **GTOL)
        got = run(DEV, A, hinted=True)
        assert rays_are_full_frame(b[f"rays_{lvl}"], Hr, Wr)
        for name, g, w_ in zip(("d_depth", "d_std", "d_volume"), got, want):
            assert_close(g, w_, name=f"{name} level {lvl} (hinted)", **GTOL)
        assert_close(got[3][:, :, :-3], want[3][:, :, :-3], name=f"d_img level {lvl} (hinted)", **GTOL)
        assert float(got[3][:, :, -3:].abs().max()) == 0.0          # the colour channels were declared data


def test_depth_bwd(enerf_fx, scatter_mode):
    from boostmvsnerfs_amd import autograd as A
    from oracle import enerf as O
    torch.manual_seed(2)
    for lvl, inv in ((0, True), (1, False)):
        prob, vals = enerf_fx.t(f"cap/cost_reg_{lvl}#0.1"), enerf_fx.t(f"cap/get_depth_values#{lvl}.0")
        g1, g2 = torch.randn(prob.shape[0], *prob.shape[2:]), torch.randn(prob.shape[0], *prob.shape[2:])
        p, v = leaf(prob), leaf(vals)
        d, s = O.depth_regress(p, v, inv)
        ((d * g1).sum() + (s * g2).sum()).backward()
        pg, vg = leaf(prob.to(DEV)), leaf(vals.to(DEV))
        d2, s2 = A.DepthRegress.apply(pg, vg, inv)
        ((d2 * g1.to(DEV)).sum() + (s2 * g2.to(DEV)).sum()).backward()
        assert_close(pg.grad, p.grad, name=f"d_prob{lvl}", **GTOL)
        assert_close(vg.grad, v.grad, name=f"d_values{lvl}", **GTOL)
    depth, std = enerf_fx.t("cap/depth_regression#0.0"), enerf_fx.t("cap/depth_regression#0.1") * 0.2
    nf0 = enerf_fx.t("cap/get_depth_values#0.1")
    g = torch.randn(1, 8, 32, 48)
    d, s = leaf(depth), leaf(std)
    (O.depth_hypotheses_cascade(d, s, nf0, 4.0, 8, True, False)[0] * g).sum().backward()
    dg, sg = leaf(depth.to(DEV)), leaf(std.to(DEV))
    (A.DepthValuesCascade.apply(dg, sg, nf0.to(DEV), 32, 48, 8)[0] * g.to(DEV)).sum().backward()
    assert_close(dg.grad, d.grad, name="cascade d_depth", **GTOL)
    assert_close(sg.grad, s.grad, name="cascade d_std", **GTOL)


def test_sweep_bwd(enerf_fx, scatter_mode):
    from boostmvsnerfs_amd import autograd as A
    from oracle import enerf as O
    torch.manual_seed(3)
    feats = {0: enerf_fx.t("cap/feature_net#0.0")[None], 1: enerf_fx.t("cap/feature_net#0.1")[None]}
    for lvl in range(2):
        P, dv = enerf_fx.t(f"cap/get_proj_mats#{lvl}"), enerf_fx.t(f"cap/get_depth_values#{lvl}.0")
        g = torch.randn(1, feats[lvl].shape[2], *dv.shape[1:])
        f, d = leaf(feats[lvl]), leaf(dv)
        (O.variance_volume(f, P, d) * g).sum().backward()
        fg, dg = leaf(feats[lvl].to(DEV)), leaf(dv.to(DEV))
        (A.SweepVariance.apply(fg, P.to(DEV), dg) * g.to(DEV)).sum().backward()
        assert_close(fg.grad, f.grad, name=f"d_feats{lvl}", **GTOL)
        assert_close(dg.grad, d.grad, name=f"d_depth_values{lvl}", rtol=5e-3, atol_scale=5e-3)
        # both kernels (default: channel-last gradient with the channel on the lane; "planar": LDS windows on the
        # reference layout) against the oracle, called directly, with and without the depth gradient
        from boostmvsnerfs_amd import ops
        for algo in ("cl", "planar"):
            for want in (True, False):
                df, ddv = ops.sweep_variance_bwd(feats[lvl].to(DEV), P.to(DEV), dv.to(DEV), g.to(DEV), want, algo=algo)
                assert_close(df, f.grad, name=f"d_feats{lvl} [{algo}]", **GTOL)
                assert (ddv is None) == (not want)
                if want:
                    assert_close(ddv, d.grad, name=f"d_depth_values{lvl} [{algo}]", rtol=5e-3, atol_scale=5e-3)
        if scatter_mode == "atomics":
            assert not ops.sweep_variance_bwd(feats[lvl].to(DEV), P.to(DEV), dv.to(DEV), g.to(DEV), False)[0].is_contiguous()   # cl ran
    # ragged: a voxel count that is not a multiple of the 64-voxel wave tile, rays leaving the source maps
    P, dv = enerf_fx.t("cap/get_proj_mats#1"), enerf_fx.t("cap/get_depth_values#1.0")
    dv2 = (dv[:, :3, :5, :7] * 1.7).contiguous()
    g2 = torch.randn(1, feats[1].shape[2], 3, 5, 7)
    f, d = leaf(feats[1]), leaf(dv2)
    (O.variance_volume(f, P, d) * g2).sum().backward()
    from boostmvsnerfs_amd import ops
    df, ddv = ops.sweep_variance_bwd(feats[1].to(DEV), P.to(DEV), dv2.to(DEV), g2.to(DEV), True, algo="cl")
    assert_close(df, f.grad, name="ragged d_feats", **GTOL)
    assert_close(ddv, d.grad, name="ragged d_depth_values", rtol=5e-3, atol_scale=5e-3)


def test_fixed_point_scatter_is_bit_reproducible_and_as_accurate(enerf_fx):
    """bmv_tuning BMV_DETERMINISTIC (csrc/scatter.hpp): every scatter gradient twice on the same inputs -> torch.equal;
    against the float-atomic form -> equal to fp32 rounding of the sums (1e-5 of the tensor's scale); NaN in -> NaN out."""
    from boostmvsnerfs_amd import _lib, ops
    torch.manual_seed(11)
    feats = enerf_fx.t("cap/feature_net#0.1")[None].to(DEV)
    P, dv = enerf_fx.t("cap/get_proj_mats#1").to(DEV), enerf_fx.t("cap/get_depth_values#1.0").to(DEV)
    g = torch.randn(1, feats.shape[2], *dv.shape[1:], device=DEV) * 1e-4        # gradients of a mean loss are small
    vol = enerf_fx.t("cap/cost_reg_1#0.0").to(DEV)
    uvd = torch.rand(1, 5000, 3, device=DEV) * 1.2 - 0.1
    go = torch.randn(1, 5000, 8, device=DEV) * 3e-6

    def run():
        a = ops.sweep_variance_bwd(feats, P, dv, g, True)
        b = ops.vox_feat_bwd(uvd, vol, go)
        return [a[0].contiguous(), a[1], b[0], b[1]]
    before = _lib.get_tuning("BMV_DETERMINISTIC")
    try:
        _lib.set_tuning("BMV_DETERMINISTIC", 0)
        ref = run()
        _lib.set_tuning("BMV_DETERMINISTIC", 1)
        one, two = run(), run()
        for x, y, r in zip(one, two, ref):
            assert torch.equal(x, y)
            scale = float(r.abs().max())
            assert float((x - r).abs().max()) <= 1e-5 * scale, (float((x - r).abs().max()), scale)
        # a launch with two outputs of different units keeps a fixed-point scale PER OUTPUT: with source maps of
        # magnitude 1e-6 the hypothesis gradient is ~1e7 times smaller than the feature gradient of the same launch; under
        # one common scale it kept 38 - 23 bits
        tiny = feats * 1e-6
        _lib.set_tuning("BMV_DETERMINISTIC", 0)
        rf, rd = ops.sweep_variance_bwd(tiny, P, dv, g, True)
        _lib.set_tuning("BMV_DETERMINISTIC", 1)
        xf, xd = ops.sweep_variance_bwd(tiny, P, dv, g, True)
        ratio = float(rf.abs().max()) / float(rd.abs().max())
        assert ratio > 1e4, ratio
        for x, r in ((xf, rf), (xd, rd)):
            assert float((x - r).abs().max()) <= 1e-5 * float(r.abs().max()), (ratio, float((x - r).abs().max()), float(r.abs().max()))
        go2 = go.clone()
        go2[0, 17, 3] = float("nan")
        assert bool(ops.vox_feat_bwd(uvd, vol, go2)[0].isnan().all())          # loud, like the float form's NaN
        assert float(ops.vox_feat_bwd(uvd, vol, torch.zeros_like(go))[0].abs().max()) == 0.0
    finally:
        _lib.set_tuning("BMV_DETERMINISTIC", before)


def test_empty_inputs_give_zero_scatter_gradients(enerf_fx, scatter_mode):
    """An empty ray shard / chunk (P = 0 samples, N = 0 rays): the scatter gradients are ZERO in both modes -- through
    the Python wrappers (an empty tensor has no device pointer: they answer without a launch) and through the C ABI
    with valid pointers, where the *_fixed entry points WRITE their outputs (the float forms add into caller-zeroed
    buffers) and used to return before writing anything."""
    import ctypes as C
    from boostmvsnerfs_amd import _lib, ops
    vol = enerf_fx.t("cap/cost_reg_1#0.0").to(DEV)
    d_vol, d_d = ops.vox_feat_bwd(torch.empty(1, 0, 3, device=DEV), vol, torch.empty(1, 0, 8, device=DEV))
    assert d_vol.shape == vol.shape and float(d_vol.abs().max()) == 0.0 and d_d.shape == (1, 0)
    img = torch.randn(1, 3, 11, 16, 24, device=DEV)
    cam = torch.eye(4, device=DEV)[None, None].repeat(1, 3, 1, 1)
    ixt = torch.eye(3, device=DEV)[None, None].repeat(1, 3, 1, 1)
    d_img, d_xyz = ops.img_feat_bwd(torch.empty(1, 0, 3, device=DEV), img, cam, ixt, cam[:, 0], 1.0,
                                    torch.empty(1, 0, 3, 11, device=DEV))
    assert d_img.shape == img.shape and float(d_img.abs().max()) == 0.0 and d_xyz.numel() == 0
    depth = torch.rand(1, 1, 8, 12, device=DEV)
    d_depth, d_std = ops.build_rays_bwd(torch.empty(1, 0, 8, device=DEV), depth, depth.clone(), torch.tensor([[2.0, 6.0]], device=DEV),
                                        torch.empty(1, 0, 2, device=DEV), 8, 12, False)
    assert float(d_depth.abs().max()) == 0.0 and float(d_std.abs().max()) == 0.0
    assert float(ops.mvs_vol_feat_bwd(torch.empty(0, 8, device=DEV), cam[0, 0], ixt[0, 0], torch.tensor([2.0, 6.0], device=DEV),
                                      torch.empty(0, 4, 8, device=DEV), 16, 24, (8, 4, 6, 8), 2).abs().max()) == 0.0
    # the C ABI with valid pointers and P = 0: the fixed-point twin overwrites a poisoned output with zeros
    lib = _lib.load()
    dummy = torch.zeros(16, device=DEV)
    out = torch.full_like(vol, float("nan")).contiguous()
    ws = ops._fixed_ws(out.numel(), vol.device)
    B, C_, D, h, w = vol.shape
    p = lambda t: C.c_void_p(t.data_ptr())       # noqa: E731
    rc = lib.bmv_vox_feat_bwd_fixed(p(dummy), p(vol.contiguous()), p(dummy), B, 0, C_, D, h, w, 0, 0, p(out), p(dummy), p(ws), _lib.stream())
    _lib.check(rc, "vox_feat_bwd_fixed")
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0
