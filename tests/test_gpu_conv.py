"""Convolution engine (csrc/conv.hip) against torch's fp32 convolutions on the same device: every
kernel shape FeatureNet / MinCostRegNet / CostRegNet use, ragged sizes (tiles that hang over the
border, W not a multiple of 16, Cin = 3, Cout in {1, 8, 9, 16, 32, 64}), fused bias / ReLU / skip,
both output layouts.  Floating-point kernel: tolerance 1e-4 of the output scale (fp32 FMA chains in a
different summation order)."""
import pytest
import torch
import torch.nn.functional as F

from boostmvsnerfs_amd import switches

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, want, tol=1e-4):
    scale = float(want.abs().max()) + 1e-12
    err = float((got - want).abs().max())
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


CASES_2D = [  # Cin, Cout, k, stride, H, W
    (3, 8, 3, 1, 37, 53), (8, 8, 3, 1, 64, 96), (8, 16, 5, 2, 64, 96), (16, 16, 3, 1, 32, 48),
    (16, 32, 5, 2, 33, 47), (32, 32, 3, 1, 16, 24), (32, 32, 1, 1, 16, 24), (16, 32, 1, 1, 19, 21),
    (32, 16, 3, 1, 32, 48), (32, 8, 3, 1, 40, 72),
]
CASES_3D = [  # Cin, Cout, stride, D, H, W
    (32, 8, 1, 8, 8, 12), (16, 8, 1, 4, 32, 48), (8, 16, 2, 8, 8, 12), (16, 16, 1, 4, 4, 6), (16, 32, 2, 4, 4, 6),
    (32, 32, 1, 2, 2, 3), (32, 64, 2, 2, 8, 12), (64, 64, 1, 1, 4, 6), (8, 9, 1, 5, 9, 19), (8, 1, 1, 8, 8, 12),
    (8, 8, 1, 3, 17, 33),
]


@pytest.mark.parametrize("Cin,Cout,k,stride,H,W", CASES_2D)
def test_conv2d(Cin, Cout, k, stride, H, W):
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(Cin * 100 + Cout)
    x = torch.randn(2, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    want = F.conv2d(x, w, b, stride, k // 2)
    wp, bp = convnet.pack_conv(w, b, stride)
    _close(convnet.conv_fwd(x, wp, bp, Cout, 1, k, stride), want)
    _close(convnet.conv_fwd(x, wp, bp, Cout, 1, k, stride, relu=True), F.relu(want))
    got = convnet.conv_fwd(x, wp, bp, Cout, 1, k, stride, channels_last=True)
    _close(got.permute(0, 3, 1, 2), want)


@pytest.mark.parametrize("Cin,Cout,stride,D,H,W", CASES_3D)
def test_conv3d(Cin, Cout, stride, D, H, W):
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(Cin * 100 + Cout + stride)
    x = torch.randn(1, Cin, D, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    want = F.conv3d(x, w, b, stride, 1)
    wp, bp = convnet.pack_conv(w, b, stride)
    _close(convnet.conv_fwd(x, wp, bp, Cout, 3, 3, stride), want)
    skip = torch.randn(want.shape, generator=g).to(DEV)
    _close(convnet.conv_fwd(x, wp, bp, Cout, 3, 3, stride, relu=True, skip=skip), F.relu(want) + skip)
    got = convnet.conv_fwd(x, wp, bp, Cout, 3, 3, stride, channels_last=True)
    _close(got.permute(0, 4, 1, 2, 3), want)


@pytest.mark.parametrize("nd,B,Cin,Cout,k,stride,sp", [
    # BASELINE configs[1] layer shapes that select the big tilings (>= 512 blocks)
    (2, 3, 3, 8, 3, 1, (512, 640)), (2, 3, 8, 16, 5, 2, (512, 640)), (2, 3, 32, 16, 3, 1, (256, 320)),
    (2, 3, 32, 32, 3, 1, (128, 160)), (2, 3, 32, 32, 1, 1, (128, 160)),
    (3, 1, 32, 8, 3, 1, (64, 64, 80)), (3, 1, 8, 16, 3, 2, (8, 256, 320)), (3, 1, 16, 16, 3, 1, (4, 128, 160)),
    (3, 1, 8, 9, 3, 1, (8, 256, 320)), (3, 1, 32, 32, 3, 1, (16, 64, 80)), (3, 1, 32, 64, 3, 2, (16, 64, 160)),
])
def test_conv_full_size_layers(nd, B, Cin, Cout, k, stride, sp):
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, *sp, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, *([k] * nd), generator=g) / (Cin * k ** nd) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    want = (F.conv3d if nd == 3 else F.conv2d)(x, w, b, stride, k // 2)
    wp, bp = convnet.pack_conv(w, b, stride)
    _close(convnet.conv_fwd(x, wp, bp, Cout, k if nd == 3 else 1, k, stride), want)


@pytest.mark.parametrize("Cin,Cout,D,H,W", [(16, 8, 4, 128, 160), (32, 16, 8, 32, 40), (64, 32, 1, 4, 6), (32, 16, 2, 8, 12), (16, 8, 4, 16, 24), (16, 8, 3, 5, 19),
                                             (32, 16, 1, 1, 1)])
def test_conv3d_transpose(Cin, Cout, D, H, W):
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(Cin + Cout + D)
    x = torch.randn(1, Cin, D, H, W, generator=g).to(DEV)
    w = (torch.randn(Cin, Cout, 3, 3, 3, generator=g) / (Cin * 27 / 8) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    want = F.conv_transpose3d(x, w, b, stride=2, padding=1, output_padding=1)
    wp, bp = convnet.pack_convT(w, b)
    _close(convnet.convT3d_fwd(x, wp, bp, Cout), want)
    skip = torch.randn(want.shape, generator=g).to(DEV)
    _close(convnet.convT3d_fwd(x, wp, bp, Cout, skip=skip), want + skip)
    _close(convnet.convT3d_fwd(x, wp, bp, Cout, relu=True), F.relu(want))


@pytest.mark.parametrize("Cf,H,W", [(8, 64, 96), (16, 32, 48), (16, 6, 70)])
def test_fpn_topdown(Cf, H, W):
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(Cf + H)
    fine = torch.randn(2, Cf, H, W, generator=g).to(DEV)
    coarse = torch.randn(2, 32, H // 2, W // 2, generator=g).to(DEV)
    w = torch.randn(32, Cf, 1, 1, generator=g).to(DEV)
    b = torch.randn(32, generator=g).to(DEV)
    want = F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True) + F.conv2d(fine, w, b)
    _close(convnet.fpn_topdown(fine, coarse, w, b), want)


@pytest.mark.parametrize("H,W,rows", [(64, 96, 8), (34, 50, 8), (6, 70, 8), (64, 96, 4), (18, 22, 4)])
def test_fpn_smooth_fused_equals_the_two_launches(H, W, rows, monkeypatch):
    """bmv_fpn_smooth_fwd = smooth0(bilinear_x2(p1) + lat0(c0)) in one launch: against torch, and against the two
    launches it replaces (same weights pack, same per-element expression for the intermediate map)."""
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(H + W)
    fine = torch.randn(2, 8, H, W, generator=g).to(DEV)
    coarse = torch.randn(2, 32, H // 2, W // 2, generator=g).to(DEV)
    wl = torch.randn(32, 8, 1, 1, generator=g).to(DEV)
    bl = torch.randn(32, generator=g).to(DEV)
    ws = (torch.randn(8, 32, 3, 3, generator=g) / (32 * 9 / 8) ** 0.5).to(DEV)
    bs = torch.randn(8, generator=g).to(DEV)
    p0 = F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True) + F.conv2d(fine, wl, bl)
    want = F.conv2d(p0, ws, bs, padding=1)
    wp, bp = convnet.pack_conv(ws, bs)
    two = convnet.conv_fwd(convnet.fpn_topdown(fine, coarse, wl, bl), wp, bp, 8, 1, 3)
    from boostmvsnerfs_amd import _lib
    _lib.set_tuning("BMV_FPN_SMOOTH_R", rows)       # (explicit library state since round 4: changeable per call)
    try:
        got = convnet.fpn_smooth(fine, coarse, wl, bl, wp, bp, 8)
    finally:
        _lib.set_tuning("BMV_FPN_SMOOTH_R", None)
    _close(got, want)
    assert float((got - two).abs().max()) <= 1e-5 * float(two.abs().max())


@pytest.mark.parametrize("H,W,rows", [(64, 96, 0), (34, 50, 9), (6, 8, 10), (18, 130, 12), (40, 124, 6), (30, 62, 8), (512, 640, 0)])
def test_fpn_smooth_s_matches_float64_like_the_fp32_kernel(H, W, rows):
    """bmv_fpn_smooth_s_fwd (csrc/fpn_s.hip): smooth0(bilinear_x2(p1) + lat0(c0)) with lat0 folded into the weights, on the
    bf16 matrix cores with three-piece fp32 operands -- against a FLOAT64 evaluation of the reference's graph
    (feature_net.py:24-36): no farther from it than the fp32 kernel it replaces (bmv_fpn_smooth_fwd), within 1e-5 of
    that kernel, not bit-equal to it; planar output and the renderer's lookup records; every row tiling; ragged strips."""
    from boostmvsnerfs_amd import _lib, convnet
    g = torch.Generator().manual_seed(H + W)
    B = 2 if H < 100 else 3
    fine = torch.randn(B, 8, H, W, generator=g).to(DEV)
    coarse = torch.randn(B, 32, H // 2, W // 2, generator=g).to(DEV)
    wl = torch.randn(32, 8, 1, 1, generator=g).to(DEV)
    bl = torch.randn(32, generator=g).to(DEV)
    ws = (torch.randn(8, 32, 3, 3, generator=g) / (32 * 9 / 8) ** 0.5).to(DEV)
    bs = torch.randn(8, generator=g).to(DEV)
    rgb = torch.rand(B, 3, H, W, generator=g).to(DEV)
    p0 = F.interpolate(coarse.double(), scale_factor=2, mode="bilinear", align_corners=True) + F.conv2d(fine.double(), wl.double(), bl.double())
    want = F.conv2d(p0, ws.double(), bs.double(), padding=1)
    ref32 = convnet.fpn_smooth(fine, coarse, wl, bl, *convnet.pack_conv(ws, bs), 8)
    _lib.set_tuning("BMV_FPN_S_ROWS", rows)
    try:
        got = convnet.fpn_smooth_s(fine, coarse, *convnet.pack_fpn_smooth_s(ws, bs, wl, bl))
        rec = convnet.fpn_smooth_s(fine, coarse, *convnet.pack_fpn_smooth_s(ws, bs, wl, bl, order=convnet.LookupRecords.EVEN_ODD), rgb=rgb)
    finally:
        _lib.set_tuning("BMV_FPN_S_ROWS", None)
    scale = float(want.abs().max())
    err, err32 = float((got.double() - want).abs().max()), float((ref32.double() - want).abs().max())
    mean, mean32 = float((got.double() - want).abs().mean()), float((ref32.double() - want).abs().mean())
    print(f"[fpn_smooth_s] {H}x{W} rows={rows}: max err {err:.3e} (fp32 kernel {err32:.3e}), mean {mean:.3e} ({mean32:.3e}), scale {scale:.3e}")
    assert err <= max(2.0 * err32, 1e-6 * scale) and mean <= 1.5 * mean32 + 1e-9 * scale
    assert float((got - ref32).abs().max()) <= 1e-5 * scale and not torch.equal(got, ref32)
    # the lookup records: [ch 0 2 4 6 | ch 1 3 5 7 | r b | g 0] per pixel, the same values as the planar map
    t = rec.t
    eo = list(convnet.LookupRecords.EVEN_ODD)
    assert torch.equal(t[..., :8].permute(0, 3, 1, 2), got[:, eo])
    assert torch.equal(t[..., 8], rgb[:, 0]) and torch.equal(t[..., 9], rgb[:, 2]) and torch.equal(t[..., 10], rgb[:, 1])
    assert float(t[..., 11].abs().max()) == 0.0


@pytest.mark.parametrize("H,W,rows", [(64, 96, 0), (34, 50, 9), (5, 7, 10), (18, 130, 12), (512, 640, 0)])
def test_conv0_s_matches_float64_like_the_fp32_kernel(H, W, rows):
    """bmv_conv0_s_fwd (csrc/fpn_s.hip): FeatureNet's first block with the second layer as bf16 MFMAs on three-piece fp32
    operands, against a FLOAT64 evaluation of the two ConvBnReLU layers: no farther from it than the fp32 fused kernel
    (bmv_conv0_fused_fwd), within 1e-5 of that kernel, not bit-equal; ragged sizes, every row tiling."""
    from boostmvsnerfs_amd import _lib, convnet
    g = torch.Generator().manual_seed(H * W)
    B = 2 if H < 100 else 3
    x = torch.randn(B, 3, H, W, generator=g).to(DEV)
    w0 = (torch.randn(8, 3, 3, 3, generator=g) / 3).to(DEV)
    b0 = torch.randn(8, generator=g).to(DEV) * 0.3
    w1 = (torch.randn(8, 8, 3, 3, generator=g) / 6).to(DEV)
    b1 = torch.randn(8, generator=g).to(DEV) * 0.3
    want = F.relu(F.conv2d(F.relu(F.conv2d(x.double(), w0.double(), b0.double(), padding=1)), w1.double(), b1.double(), padding=1))
    ref32 = convnet.conv0_fused(x, w0.contiguous(), b0.contiguous(), *convnet.pack_conv(w1, b1), 8)
    _lib.set_tuning("BMV_CONV0_S_ROWS", rows)
    try:
        got = convnet.conv0_s(x, *convnet.pack_conv0_s(w0, b0, w1, b1))
    finally:
        _lib.set_tuning("BMV_CONV0_S_ROWS", None)
    scale = float(want.abs().max())
    err, err32 = float((got.double() - want).abs().max()), float((ref32.double() - want).abs().max())
    mean, mean32 = float((got.double() - want).abs().mean()), float((ref32.double() - want).abs().mean())
    print(f"[conv0_s] {H}x{W} rows={rows}: max err {err:.3e} (fp32 kernel {err32:.3e}), mean {mean:.3e} ({mean32:.3e}), scale {scale:.3e}")
    assert err <= max(2.0 * err32, 1e-6 * scale) and mean <= 1.5 * mean32 + 1e-9 * scale
    assert float((got - ref32).abs().max()) <= 1e-5 * scale and not torch.equal(got, ref32)


C2S_CASES = [(8, 16, 5, 2, 2, 64, 96, 0), (16, 32, 5, 2, 2, 64, 96, 0), (16, 16, 3, 1, 2, 32, 48, 0), (32, 32, 3, 1, 2, 32, 48, 0),
             (8, 16, 5, 2, 1, 34, 50, 8), (16, 32, 5, 2, 1, 6, 70, 4), (16, 16, 3, 1, 1, 18, 130, 8), (32, 32, 3, 1, 1, 2, 2, 4),
             (8, 32, 5, 2, 1, 20, 36, 8), (32, 16, 3, 1, 1, 10, 34, 4),
             (8, 16, 5, 2, 3, 512, 640, 0), (16, 16, 3, 1, 3, 256, 320, 0), (16, 32, 5, 2, 3, 256, 320, 0),
             (32, 32, 3, 1, 3, 128, 160, 0)]


@pytest.mark.parametrize("Cin,Cout,ks,stride,B,H,W,rows", C2S_CASES)
def test_conv2d_s_matches_float64_like_the_fp32_kernel(Cin, Cout, ks, stride, B, H, W, rows):
    """bmv_conv2d_s_fwd (csrc/conv2d_s.hip): FeatureNet's encoder layers as bf16 MFMAs on three-piece fp32 operands, against
    a FLOAT64 convolution: no farther from it than the fp32 engine (bmv_conv_fwd), within 1e-5 of that kernel, not
    bit-equal; ragged sizes (strips and row blocks that end inside a tile, a 2 x 2 image), both row tilings, the frame's
    own layer shapes."""
    from boostmvsnerfs_amd import _lib, convnet
    g = torch.Generator().manual_seed(Cin * 1000 + H * W)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=ks // 2))
    ref32 = convnet.conv_fwd(x, *convnet.pack_conv(w, b, stride=stride), Cout, 1, ks, stride, relu=True)
    _lib.set_tuning("BMV_CONV2D_S_ROWS", rows)
    try:
        got = convnet.conv2d_s(x, *convnet.pack_conv2d_s(w, b), Cout, ks, stride, relu=True)
    finally:
        _lib.set_tuning("BMV_CONV2D_S_ROWS", None)
    assert got.shape == want.shape
    scale = float(want.abs().max())
    err, err32 = float((got.double() - want).abs().max()), float((ref32.double() - want).abs().max())
    mean, mean32 = float((got.double() - want).abs().mean()), float((ref32.double() - want).abs().mean())
    print(f"[conv2d_s] {Cin}->{Cout} k{ks}s{stride} {B}x{H}x{W} rows={rows}: max err {err:.3e} (fp32 kernel {err32:.3e}), "
          f"mean {mean:.3e} ({mean32:.3e}), scale {scale:.3e}")
    assert err <= max(2.0 * err32, 1e-6 * scale) and mean <= 1.5 * mean32 + 1e-9 * scale
    assert float((got - ref32).abs().max()) <= 1e-5 * scale and not torch.equal(got, ref32)
    # leaky slope, no bias
    got = convnet.conv2d_s(x, *convnet.pack_conv2d_s(w, None), Cout, ks, stride, slope=0.1)
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), None, stride=stride, padding=ks // 2), 0.1)
    assert float((got.double() - want).abs().max()) <= 4e-6 * scale


def test_conv2d_s_refuses_what_it_is_not_built_for():
    """Shapes outside FeatureNet's encoder are errors by name, not another kernel: odd sizes, other channel counts, a 3x3
    stride-2 layer; and the entry point wants exactly one input form."""
    from boostmvsnerfs_amd import _lib, convnet
    lib = _lib.load()
    assert lib.bmv_conv2d_s_wsplit_ints(24, 16, 3, 1) == 0 and lib.bmv_conv2d_s_wsplit_ints(16, 16, 3, 2) == 0
    assert lib.bmv_conv2d_s_wsplit_ints(8, 16, 3, 1) == 0 and lib.bmv_conv2d_s_wsplit_ints(16, 24, 5, 2) == 0
    w = torch.randn(16, 16, 3, 3, device=DEV)
    pk = convnet.pack_conv2d_s(w, None)
    with pytest.raises(RuntimeError, match="not covered"):
        convnet.conv2d_s(torch.randn(1, 16, 33, 48, device=DEV), *pk, 16, 3, 1)
    with pytest.raises(AssertionError):
        convnet.pack_conv2d_s(torch.randn(16, 24, 3, 3, device=DEV), None)
    x = torch.randn(1, 16, 8, 8, device=DEV)
    with pytest.raises(RuntimeError, match="exactly one of"):
        _lib.check(lib.bmv_conv2d_s_fwd(_lib.dptr(x), _lib.dptr(convnet.SplitRecords.from_planar(x).data, "r", torch.int32),
                                        _lib.dptr(pk[0], "w", torch.int32), _lib.dptr(pk[1]), _lib.dptr(torch.empty(1, 16, 8, 8, device=DEV)),
                                        None, 1, 16, 8, 8, 16, 3, 1, 1.0, _lib.stream()), "conv2d_s_fwd")


@pytest.mark.parametrize("Cin,Cout,ks,stride,B,H,W,rows", [c for c in C2S_CASES if c[4] < 3 or c[0] != 32])
def test_conv2d_s_split_records_are_bit_identical_to_the_planar_path(Cin, Cout, ks, stride, B, H, W, rows):
    """The split-record form of a map between two layers (convnet.SplitRecords: written by the producer's epilogue, staged by
    the consumer with LDS-DMA) changes WHERE a value is split, not the arithmetic: records in -> the planar result bit for
    bit; records out -> exactly the records of the planar result; both at once; ragged sizes (image borders = the zero
    record, rows outside the image requested and never multiplied), both row tilings."""
    from boostmvsnerfs_amd import _lib, convnet
    g = torch.Generator().manual_seed(Cin * 1000 + H * W + 1)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    pk = convnet.pack_conv2d_s(w, b)
    _lib.set_tuning("BMV_CONV2D_S_ROWS", rows)
    try:
        want = convnet.conv2d_s(x, *pk, Cout, ks, stride, relu=True)
        xr = convnet.SplitRecords.from_planar(x)
        assert torch.equal(xr.to_planar(), x)
        got = convnet.conv2d_s(xr, *pk, Cout, ks, stride, relu=True)
        assert torch.equal(got, want)
        rec = convnet.conv2d_s(x, *pk, Cout, ks, stride, relu=True, records=True)
        assert torch.equal(rec.data, convnet.SplitRecords.from_planar(want).data) and torch.equal(rec.to_planar(), want)
        both, rec2 = convnet.conv2d_s(xr, *pk, Cout, ks, stride, relu=True, records="both")
        assert torch.equal(both, want) and torch.equal(rec2.data, rec.data)
    finally:
        _lib.set_tuning("BMV_CONV2D_S_ROWS", None)


@pytest.mark.parametrize("H,W", [(64, 96), (34, 50), (5, 7), (18, 130)])
def test_conv0_fused_equals_the_two_launches(H, W):
    """bmv_conv0_fused_fwd = relu(conv(relu(conv(x)))) of FeatureNet's first block in one launch, against torch and
    against the two engine launches it replaces."""
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(2, 3, H, W, generator=g).to(DEV)
    w0 = (torch.randn(8, 3, 3, 3, generator=g) / 3).to(DEV)
    b0 = torch.randn(8, generator=g).to(DEV) * 0.3
    w1 = (torch.randn(8, 8, 3, 3, generator=g) / 6).to(DEV)
    b1 = torch.randn(8, generator=g).to(DEV) * 0.3
    want = F.relu(F.conv2d(F.relu(F.conv2d(x, w0, b0, padding=1)), w1, b1, padding=1))
    wp0, bp0 = convnet.pack_conv(w0, b0)
    wp1, bp1 = convnet.pack_conv(w1, b1)
    two = convnet.conv_fwd(convnet.conv_fwd(x, wp0, bp0, 8, 1, 3, relu=True), wp1, bp1, 8, 1, 3, relu=True)
    got = convnet.conv0_fused(x, w0.contiguous(), b0.contiguous(), wp1, bp1, 8)
    _close(got, want)
    assert float((got - two).abs().max()) <= 1e-5 * float(two.abs().max())


@pytest.mark.parametrize("H,W", [(32, 48), (17, 35), (8, 16), (3, 5)])
def test_conv_top_fused_equals_the_two_launches(H, W):
    """bmv_conv_top_fwd = conv1x1(relu(conv3x3(x))) with 32 channels throughout, channel-last output: against torch and
    against the two engine launches it replaces."""
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(H * W + 1)
    x = torch.randn(2, 32, H, W, generator=g).to(DEV)
    w1 = (torch.randn(32, 32, 3, 3, generator=g) / 12).to(DEV)
    b1 = torch.randn(32, generator=g).to(DEV) * 0.3
    w2 = (torch.randn(32, 32, 1, 1, generator=g) / 4).to(DEV)
    b2 = torch.randn(32, generator=g).to(DEV) * 0.3
    want = F.conv2d(F.relu(F.conv2d(x, w1, b1, padding=1)), w2, b2).permute(0, 2, 3, 1)
    wp1, bp1 = convnet.pack_conv(w1, b1)
    wp2, bp2 = convnet.pack_conv(w2, b2)
    two = convnet.conv_fwd(convnet.conv_fwd(x, wp1, bp1, 32, 1, 3, relu=True), wp2, bp2, 32, 1, 1, channels_last=True)
    got = convnet.conv_top(x, wp1, bp1, wp2, bp2)
    _close(got, want)
    assert float((got - two).abs().max()) <= 1e-5 * float(two.abs().max())


def _randomise_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) or hasattr(m, "running_var"):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def test_feature_net_engine_matches_torch_modules(monkeypatch):
    from boostmvsnerfs_amd.networks.enerf.cnn import FeatureNet
    torch.manual_seed(0)
    net = FeatureNet()
    _randomise_bn(net, 1)
    net = net.to(DEV).eval()
    x = torch.randn(3, 3, 64, 96, device=DEV)
    with torch.no_grad():
        got = net(x)
        monkeypatch.setitem(switches.VALUES, "BMV_CNN", "torch")
        want = net(x)
    from boostmvsnerfs_amd.ops import QuadFeats
    assert isinstance(got[0], QuadFeats) and isinstance(got[1], QuadFeats)                  # the sweep's layout
    for g_, w_ in zip(got, want):
        assert g_.shape == w_.shape
        _close(g_.contiguous(), w_)
    # ... and the channel-last form (training / other sweep kernels): (N,C,H,W) views of (N,H,W,C) buffers
    monkeypatch.delitem(switches.VALUES, "BMV_CNN")
    net.quad_out = False
    with torch.no_grad():
        got_cl = net(x)
    net.quad_out = True
    assert not got_cl[0].is_contiguous() and got_cl[0].permute(0, 2, 3, 1).is_contiguous()
    for g_, w_ in zip(got_cl, want):
        _close(g_, w_)
    monkeypatch.setitem(switches.VALUES, "BMV_CNN", "torch")
    # a parameter update invalidates the packed weights
    monkeypatch.delitem(switches.VALUES, "BMV_CNN")
    with torch.no_grad():
        net.smooth0.bias.add_(1.0)
        _close(net(x)[2], want[2] + 1.0)


def test_feature_net_at_frame_size_bf16_kernels_on_and_off(monkeypatch):
    """At BASELINE configs[1]'s size (3 x 512 x 640) FeatureNet's default forward runs its first block, the encoder's 5x5
    stride-2 / 3x3 layers and the last top-down step on the bf16 matrix cores (BMV_CONV0_S, BMV_CONV2D_S, BMV_FPN_S:
    three-piece fp32 operands; the small fixtures of the other tests are below the size gate).  Both forms against the
    torch modules, and against each other to fp32 rounding -- not bit-equal, and the switch counts its launches."""
    from boostmvsnerfs_amd import convnet
    from boostmvsnerfs_amd.networks.enerf.cnn import FeatureNet
    torch.manual_seed(0)
    net = FeatureNet()
    _randomise_bn(net, 1)
    net = net.to(DEV).eval()
    x = torch.rand(3, 3, 512, 640, device=DEV)
    calls = []
    real = convnet.conv2d_s

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    monkeypatch.setattr(convnet, "conv2d_s", spy)
    with torch.no_grad():
        got_s = [t.contiguous() if torch.is_tensor(t) else t.to_nchw() for t in net(x)]
        assert len(calls) == 3, calls                           # conv1.0, conv1.1, conv2.0
        monkeypatch.setitem(switches.VALUES, "BMV_CONV2D_S_REC", "0")     # planar maps between the layers: the same bits
        got_p = [t.contiguous() if torch.is_tensor(t) else t.to_nchw() for t in net(x)]
        assert len(calls) == 6 and all(torch.equal(a, b_) for a, b_ in zip(got_s, got_p))
        for name in ("BMV_CONV0_S", "BMV_CONV2D_S", "BMV_FPN_S"):
            monkeypatch.setitem(switches.VALUES, name, "0")
        got_f = [t.contiguous() if torch.is_tensor(t) else t.to_nchw() for t in net(x)]
        assert len(calls) == 6
        monkeypatch.setitem(switches.VALUES, "BMV_CNN", "torch")
        want = net(x)
    for a, b_, w_ in zip(got_s, got_f, want):
        _close(a, w_), _close(b_, w_)
        scale = float(w_.abs().max())
        assert float((a - b_).abs().max()) <= 1e-5 * scale and not torch.equal(a, b_)


@pytest.mark.parametrize("cls,cin,shape", [("MinCostRegNet", 32, (8, 8, 12)), ("CostRegNet", 16, (8, 32, 48))])
def test_cost_reg_engine_matches_torch_modules(monkeypatch, cls, cin, shape):
    from boostmvsnerfs_amd.networks.enerf import cnn
    torch.manual_seed(0)
    net = getattr(cnn, cls)(cin)
    _randomise_bn(net, 2)
    net = net.to(DEV).eval()
    x = torch.rand(1, cin, *shape, device=DEV)
    with torch.no_grad():
        feat, prob = net(x)
        monkeypatch.setitem(switches.VALUES, "BMV_CNN", "torch")
        feat_t, prob_t = net(x)
    assert feat.shape == feat_t.shape and prob.shape == prob_t.shape
    _close(feat, feat_t)
    _close(prob, prob_t)


def test_leaky_slope_and_mvsnerf_stacks(monkeypatch):
    """InPlaceABN's leaky ReLU (slope 0.01) as the epilogue; MVSNeRF's FeatureNet / CostRegNet
    (mvsnerf/network.py:699-779: 41 input channels, three stride-2 levels) against their torch modules."""
    from boostmvsnerfs_amd import convnet
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    set_cfg(make_cfg("mvsnerf_eval"))
    from boostmvsnerfs_amd.networks.mvsnerf import network as M
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 8, 4, 9, 21, generator=g).to(DEV)
    w = (torch.randn(16, 8, 3, 3, 3, generator=g) / 15).to(DEV)
    b = torch.randn(16, generator=g).to(DEV)
    wp, bp = convnet.pack_conv(w, b)
    _close(convnet.conv_fwd(x, wp, bp, 16, 3, 3, slope=0.01), F.leaky_relu(F.conv3d(x, w, b, 1, 1), 0.01))
    torch.manual_seed(0)
    feat, reg = M.FeatureNet(), M.CostRegNet(41)
    _randomise_bn(feat, 3), _randomise_bn(reg, 4)
    feat, reg = feat.to(DEV).eval(), reg.to(DEV).eval()
    img = torch.randn(1, 3, 3, 64, 96, device=DEV)
    vol = torch.rand(1, 41, 8, 40, 48, device=DEV)
    with torch.no_grad():
        got_f, got_r = feat(img), reg(vol)
        monkeypatch.setitem(switches.VALUES, "BMV_CNN", "torch")
        want_f, want_r = feat(img), reg(vol)
    assert got_f.shape == want_f.shape and got_r.shape == want_r.shape
    _close(got_f, want_f)
    _close(got_r, want_r)


def test_fold_bn_matches_eval_batch_norm():
    from boostmvsnerfs_amd import convnet
    torch.manual_seed(0)
    conv = torch.nn.Conv3d(8, 16, 3, padding=1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm3d(16).to(DEV).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.normal_(), bn.running_mean.normal_(), bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(1, 8, 4, 8, 12, device=DEV)
        want = F.relu(bn(conv(x)))
        wp, bp = convnet.pack_conv(*convnet.fold_bn(conv.weight, bn))
        _close(convnet.conv_fwd(x, wp, bp, 16, 3, 3, relu=True), want)


def test_unsupported_shape_is_loud():
    from boostmvsnerfs_amd import convnet
    x = torch.zeros(1, 4, 8, 8, device=DEV)
    wp, bp = convnet.pack_conv(torch.zeros(4, 4, 7, 7, device=DEV), None)
    with pytest.raises(RuntimeError, match="not one of the shapes"):
        convnet.conv_fwd(x, wp, bp, 4, 1, 7, 1)


# ---------------------------------------------------------------------------------------------------------------
# training leg: weight gradient of the 3x3x3 (transposed) convolutions on the MFMA kernel (csrc/conv_wgrad.hip)
# vs torch autograd in float64 (lib/networks/enerf/cost_reg_net.py:4-86)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,stride,dhw", [
    (16, 8, 1, (3, 9, 70)),       # level-1 conv0 shape class: tail chunk of 6 voxels
    (8, 16, 2, (4, 10, 20)),      # strided, one partial chunk
    (8, 16, 2, (5, 7, 9)),        # odd sizes
    (32, 64, 2, (4, 6, 10)),      # two output blocks
    (64, 64, 1, (2, 4, 5)),       # four block pairs, row shorter than a float4 pair
    (8, 1, 1, (2, 5, 130)),       # depth head: one output channel, three chunks
    (16, 8, 1, (8, 128, 160)),    # 10 240 voxel groups: every wave walks its grid-stride loop more than once
])
def test_conv3d_weight_gradient(cin, cout, stride, dhw):
    from boostmvsnerfs_amd.networks.enerf.conv_train import Conv3d
    torch.manual_seed(0)
    m = Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False).to(DEV)
    big = dhw[0] * dhw[1] * dhw[2] > 100000
    x = torch.randn(1 if big else 2, cin, *dhw, device=DEV, requires_grad=True)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    ref_dev = "cpu" if big else DEV          # float64 reference (CPU for the large case: no fp64 GPU convolution needed)
    xd = x.detach().double().to(ref_dev).requires_grad_(True)
    wd = m.weight.detach().double().to(ref_dev).requires_grad_(True)
    torch.nn.functional.conv3d(xd, wd, None, stride, 1).backward(gy.double().to(ref_dev))
    gw, gx = wd.grad.to(DEV), xd.grad.to(DEV)
    yd = torch.nn.functional.conv3d(xd.detach(), wd.detach(), None, stride, 1).to(DEV)
    assert float((y.double() - yd).abs().max()) <= 2e-5 * float(yd.abs().max())        # engine forward (repacked weights)
    assert float((m.weight.grad.double() - gw).abs().max()) <= 2e-5 * float(gw.abs().max())
    assert float((x.grad.double() - gx).abs().max()) <= 1e-4 * float(gx.abs().max())


@pytest.mark.parametrize("cin,cout,dhw", [(64, 32, (2, 4, 5)), (16, 8, (4, 9, 40)), (32, 16, (3, 5, 33))])
def test_conv_transpose3d_weight_gradient(cin, cout, dhw):
    from boostmvsnerfs_amd.networks.enerf.conv_train import ConvTranspose3d
    torch.manual_seed(1)
    m = ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False).to(DEV)
    x = torch.randn(1, cin, *dhw, device=DEV, requires_grad=True)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    wd = m.weight.detach().double().requires_grad_(True)
    torch.nn.functional.conv_transpose3d(xd, wd, None, stride=2, padding=1, output_padding=1).backward(gy.double())
    scale = float(wd.grad.abs().max())
    assert float((m.weight.grad.double() - wd.grad).abs().max()) <= 2e-5 * scale
    assert float((x.grad.double() - xd.grad).abs().max()) <= 1e-4 * float(xd.grad.abs().max())


# ---------------------------------------------------------------------------------------------------------------
# training-mode batch norm (+ ReLU) on csrc/bn.hip vs torch in float64 (lib/networks/enerf/utils.py:10-33)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,relu", [((3, 8, 37, 53), True), ((1, 16, 4, 32, 48), True), ((1, 32, 2, 4, 5), False),
                                         ((2, 8, 3, 17, 33), False), ((2, 64, 1, 2, 3), True), ((1, 8, 8, 256, 320), True),
                                         ((2, 16, 3, 9, 20), 0.01)])       # InPlaceABN's leaky ReLU (MVSNeRF stacks)
def test_batch_norm_training(shape, relu):
    from boostmvsnerfs_amd import autograd as A
    torch.manual_seed(0)
    C = shape[1]
    x = (torch.randn(shape, device=DEV) * 3 + 50.0)                     # a mean far from zero: nothing may cancel against it
    x.requires_grad_(True)
    w = torch.rand(C, device=DEV, requires_grad=True)
    b = torch.randn(C, device=DEV, requires_grad=True)
    rm, rv = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
    rm_d, rv_d = rm.double().clone(), rv.double().clone()
    y = A.BatchNormTrain.apply(x, w, b, rm, rv, 1e-5, 0.1, relu)
    gy = torch.randn_like(y)
    y.backward(gy)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = F.batch_norm(xd, rm_d, rv_d, wd, bd, True, 0.1, 1e-5)
    if relu is not False:
        # the mask of the fp32 output: values within rounding of zero may fall on either side, and every flip moves
        # the channel sums by a whole term -- both sides must differentiate the same function
        slope = 0.0 if relu is True else float(relu)
        assert float((F.leaky_relu(yd, slope) - y.double()).abs().max()) <= 2e-5 * float(yd.abs().max())
        yd = yd * torch.where(y.detach() > 0, 1.0, slope).double()
    yd.backward(gy.double())
    _close(y.double(), yd.detach(), 2e-5)
    _close(rm.double(), rm_d, 1e-5), _close(rv.double(), rv_d, 1e-4)
    _close(x.grad.double(), xd.grad, 2e-4)
    _close(w.grad.double(), wd.grad, 2e-4), _close(b.grad.double(), bd.grad, 2e-4)


def test_cost_reg_training_forward_matches_torch_modules(monkeypatch):
    """The whole training-mode regulariser (our batch norm, merged heads) == the same modules on torch's batch norm."""
    from boostmvsnerfs_amd.networks.enerf.cnn import MinCostRegNet
    torch.manual_seed(0)
    net = MinCostRegNet(16).to(DEV).train()
    ref = MinCostRegNet(16).to(DEV).train()
    ref.load_state_dict(net.state_dict())
    x = torch.randn(1, 16, 8, 32, 48, device=DEV)
    f1, d1 = net(x)
    monkeypatch.setitem(switches.VALUES, "BMV_BN", "torch")
    f2, d2 = ref(x)
    _close(f1, f2, 2e-4), _close(d1, d2, 2e-4)
    for (k, a), (_, bb) in zip(net.state_dict().items(), ref.state_dict().items()):
        _close(a.float(), bb.float(), 2e-4) if a.dtype.is_floating_point else None
        assert a.dtype.is_floating_point or torch.equal(a, bb), k


@pytest.mark.parametrize("cin,cout,k,stride,bias,hw", [(3, 8, 3, 1, False, (37, 53)), (8, 16, 5, 2, False, (64, 96)),
                                                        (32, 32, 1, 1, True, (16, 24)), (32, 8, 3, 1, True, (40, 72)),
                                                        (16, 32, 1, 1, True, (19, 21)), (16, 32, 5, 2, False, (32, 48)),
                                                        (8, 16, 5, 2, False, (63, 95))])
def test_conv2d_training_module(cin, cout, k, stride, bias, hw):
    """FeatureNet's convolutions under autograd: forward, data gradient and weight gradient on the engine (the 5x5
    stride-2 data gradient as one 3x3 convolution over the four input parities; odd sizes: aten), vs float64."""
    from boostmvsnerfs_amd.networks.enerf.conv_train import Conv2d
    torch.manual_seed(2)
    m = Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=bias).to(DEV)
    x = torch.randn(3, cin, *hw, device=DEV, requires_grad=True)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    wd = m.weight.detach().double().requires_grad_(True)
    bd = m.bias.detach().double().requires_grad_(True) if bias else None
    yd = F.conv2d(xd, wd, bd, stride, k // 2)
    yd.backward(gy.double())
    _close(y.double(), yd.detach(), 2e-5)
    _close(x.grad.double(), xd.grad, 1e-4), _close(m.weight.grad.double(), wd.grad, 1e-4)
    if bias:
        _close(m.bias.grad.double(), bd.grad, 1e-4)


def test_weight_pack_cache_is_per_tensor_object_and_version():
    """convnet.pack_conv_dev keeps one device-side pack per weight and version.  A tensor that went away must not lend
    its pack to a new one the caching allocator puts at the same address (both with version counter 0: the temporaries
    of the 5x5 stride-2 data gradient, the parameters of the next network of a process), an in-place update must
    replace the pack, and an unchanged parameter must be packed once."""
    from boostmvsnerfs_amd import convnet
    convnet.clear_pack_cache()
    torch.manual_seed(3)
    x = torch.randn(1, 8, 24, 40, device=DEV)
    ptrs = set()
    for seed in range(4):                          # temporaries: same size, freed before the next one is made
        w = torch.randn(8, 8, 3, 3, device=DEV, generator=torch.Generator(DEV).manual_seed(seed))
        ptrs.add(w.data_ptr())
        y = convnet.conv_fwd(x, *convnet.pack_conv_dev(w, None, 1), 8, 1, 3, 1)
        _close(y.double(), F.conv2d(x.double(), w.double(), None, 1, 1), 2e-5)
        del w, y
    assert len(ptrs) < 4                           # (the allocator did hand an address out again: the case is exercised)
    w = torch.randn(8, 8, 3, 3, device=DEV)
    p1, _ = convnet.pack_conv_dev(w, None, 1)
    p2, _ = convnet.pack_conv_dev(w, None, 1)
    assert p1 is p2                                # unchanged: one pack
    w.mul_(2.0)                                    # (what an optimiser step does: in place, version + 1)
    p3, b3 = convnet.pack_conv_dev(w, None, 1)
    assert p3 is not p1
    _close(convnet.conv_fwd(x, p3, b3, 8, 1, 3, 1).double(), F.conv2d(x.double(), w.double(), None, 1, 1), 2e-5)
    convnet.clear_pack_cache()


@pytest.mark.parametrize("parts", [3, 2])
@pytest.mark.parametrize("cin,cout,dhw", [(16, 8, (8, 40, 72)), (32, 8, (5, 19, 52)), (8, 9, (3, 16, 36)), (16, 16, (2, 8, 32))])
def test_conv3d_split_bf16(cin, cout, dhw, parts):
    """csrc/conv_split.hip: 3x3x3 convolution on the bf16 matrix cores with split fp32 operands, fp32 accumulation, against
    float64.  parts = 3 (hi + mid + lo: the operand exactly; six MFMAs per product group) must be as close to float64 as
    the fp32 engine is; parts = 2 (three MFMAs; the opt-in experiment) within 2^-16-class error."""
    from boostmvsnerfs_amd import convnet
    torch.manual_seed(4)
    x = torch.randn(2, cin, *dhw, device=DEV)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cin) ** 0.5
    b = torch.randn(cout, device=DEV)
    y = convnet.conv3d_split_fwd(x, *convnet.pack_conv_split(w, b, parts=parts), cout, relu=True)
    yd = F.relu(F.conv3d(x.double(), w.double(), b.double(), 1, 1))
    y32 = convnet.conv_fwd(x, *convnet.pack_conv(w, b), cout, 3, 3, relu=True)
    scale = float(yd.abs().max())
    err, err32 = float((y.double() - yd).abs().max()), float((y32.double() - yd).abs().max())
    print(f"[conv_split{parts}] {cin}->{cout} {dhw}: max err {err:.3e}, fp32 engine {err32:.3e}, scale {scale:.3e}")
    if parts == 3:
        assert err <= max(2.0 * err32, 1e-6 * scale)          # fp32-equivalent
    else:
        assert err <= 2e-5 * scale


C4_CASES = [  # nd, B, Cin, Cout, spatial: ragged tiles, Cin not a multiple of 4, every cout-group count, both ranks
    (3, 1, 32, 8, (8, 8, 12)), (3, 1, 16, 8, (4, 32, 48)), (3, 2, 8, 9, (5, 9, 19)), (3, 1, 8, 8, (3, 17, 33)),
    (3, 1, 6, 4, (2, 5, 7)), (3, 1, 8, 12, (4, 16, 16)), (3, 1, 8, 1, (7, 20, 18)),
    (2, 2, 32, 8, (40, 72)), (2, 3, 16, 8, (33, 47)), (2, 1, 3, 8, (37, 53)),
    # BASELINE configs[1] shapes: the regularisers' first layers and heads, FeatureNet's smooth0
    (3, 1, 32, 8, (64, 64, 80)), (3, 1, 16, 8, (8, 256, 320)), (3, 1, 8, 9, (8, 256, 320)), (2, 3, 32, 8, (512, 640)),
]


@pytest.mark.parametrize("nd,B,Cin,Cout,sp", C4_CASES)
def test_conv_c4(nd, B, Cin, Cout, sp):
    """csrc/conv_c4.hip (few output channels on the 4 x 4 x 1 matrix blocks) against torch's fp32 convolution and against
    the 16-row engine kernel it replaces; every tuning variant; bias / ReLU / leaky slope; the volume-record epilogue."""
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(Cin * 100 + Cout + nd)
    x = torch.randn(B, Cin, *sp, generator=g).to(DEV)
    k3 = (3,) * nd
    w = (torch.randn(Cout, Cin, *k3, generator=g) / (Cin * 3 ** nd) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    conv = F.conv3d if nd == 3 else F.conv2d
    want = conv(x, w, b, 1, 1)
    wp, bp = convnet.pack_conv_c4(w, b)
    big = x.numel() > 4e6
    for variant in ((0,) if big else ((0, 1, 2, 4) if nd == 3 else (0, 2))):
        if variant == 4 and Cout not in (8, 9, 12):
            continue
        _close(convnet.conv_c4_fwd(x, wp, bp, Cout, variant=variant), want, tol=2e-5)
    _close(convnet.conv_c4_fwd(x, wp, bp, Cout, relu=True), F.relu(want), tol=2e-5)
    _close(convnet.conv_c4_fwd(x, wp, bp, Cout, slope=0.01), F.leaky_relu(want, 0.01), tol=2e-5)
    wp16, bp16 = convnet.pack_conv(w, b, 1)
    _close(convnet.conv_c4_fwd(x, wp, bp, Cout), convnet.conv_fwd(x, wp16, bp16, Cout, 3 if nd == 3 else 1, 3, 1), tol=2e-5)
    if nd == 3 and Cin % 4 == 0:
        # the input as quad records (the plane sweep's output layout): bit-identical to the planar input
        from boostmvsnerfs_amd import ops
        D_, H_, W_ = sp
        qv = ops.QuadVolume(x.view(B, Cin // 4, 4, D_, H_, W_).permute(0, 1, 3, 4, 5, 2).contiguous())
        assert torch.equal(qv.to_planar(), x)
        assert torch.equal(convnet.conv_c4_fwd(qv, wp, bp, Cout, relu=True), convnet.conv_c4_fwd(x, wp, bp, Cout, relu=True))
        if Cout in (8, 9):
            (r0, l0), (r1, l1) = convnet.conv_c4_fwd(qv, wp, bp, Cout, records=True), convnet.conv_c4_fwd(x, wp, bp, Cout, records=True)
            assert torch.equal(r0.t, r1.t) and (l0 is None or torch.equal(l0, l1))
        if Cout % 4 == 0:     # ... and the result as quad records (for the stride-2 layer and conv11's skip add)
            qo = convnet.conv_c4_fwd(qv, wp, bp, Cout, relu=True, quad_out=True)
            assert isinstance(qo, ops.QuadVolume) and torch.equal(qo.to_planar(), convnet.conv_c4_fwd(x, wp, bp, Cout, relu=True))
    if nd == 3 and Cout in (8, 9):
        rec, logit = convnet.conv_c4_fwd(x, wp, bp, Cout, records=True)
        _close(rec.t.permute(0, 4, 1, 2, 3), want[:, :8], tol=2e-5)          # (channels in the order they were packed)
        if Cout == 9:
            _close(logit, want[:, 8], tol=2e-5)


C4S_CASES = [  # B, Cin, Cout, pair, spatial: ragged tiles in every axis, one plane, the frame's four layers at full size
    (1, 16, 8, True, (4, 32, 48)), (2, 32, 8, True, (3, 17, 33)), (1, 8, 9, False, (5, 9, 19)), (1, 8, 8, False, (2, 16, 16)),
    (1, 8, 8, True, (1, 5, 7)), (1, 16, 12, False, (2, 20, 18)), (1, 24, 16, False, (2, 7, 40)),
    (1, 32, 8, True, (64, 64, 80)), (1, 16, 8, True, (8, 256, 320)), (1, 8, 9, False, (8, 256, 320)), (1, 8, 9, False, (64, 64, 80)),
]


@pytest.mark.parametrize("B,Cin,Cout,pair,sp", C4S_CASES)
def test_conv_c4s(B, Cin, Cout, pair, sp):
    """csrc/conv_c4s.hip: the regularisers' first layers / heads on the bf16 matrix cores with three-piece fp32 operands,
    staged from quad records, against a FLOAT64 convolution: no farther from it than the fp32 kernel it replaces
    (csrc/conv_c4.hip) -- the claim 'fp32 accuracy' as a test -- and not bit-equal to that kernel (the split path ran);
    bias / ReLU / leaky slope; the three output forms (planar, quad records, the renderer's volume records) bit-equal."""
    from boostmvsnerfs_amd import convnet, ops
    g = torch.Generator().manual_seed(Cin * 100 + Cout + sp[0])
    x = torch.randn(B, Cin, *sp, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    D_, H_, W_ = sp
    qv = ops.QuadVolume(x.view(B, Cin // 4, 4, D_, H_, W_).permute(0, 1, 3, 4, 5, 2).contiguous())
    ws, bs, pr = convnet.pack_conv_c4s(w, b, pair)
    assert pr == pair
    want64 = F.conv3d(x.double(), w.double(), b.double(), 1, 1)
    got = convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout)
    if Cout <= 12:
        ref32 = convnet.conv_c4_fwd(qv, *convnet.pack_conv_c4(w, b), Cout)
    else:
        ref32 = convnet.conv_fwd(x, *convnet.pack_conv(w, b, 1), Cout, 3, 3, 1)
    scale = float(want64.abs().max())
    err, err32 = float((got.double() - want64).abs().max()), float((ref32.double() - want64).abs().max())
    mean, mean32 = float((got.double() - want64).abs().mean()), float((ref32.double() - want64).abs().mean())
    print(f"[conv_c4s] {Cin}->{Cout} pair={pair} {sp}: max err {err:.3e} (fp32 kernel {err32:.3e}), mean {mean:.3e} ({mean32:.3e}), scale {scale:.3e}")
    assert err <= max(2.0 * err32, 1e-6 * scale) and mean <= 1.5 * mean32 + 1e-9 * scale
    assert not torch.equal(got, ref32), "the split path did not run"
    _close(convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, relu=True), F.relu(want64).float(), tol=2e-6)
    _close(convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, slope=0.01), F.leaky_relu(want64, 0.01).float(), tol=2e-6)
    if Cout % 4 == 0:
        qo = convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, relu=True, quad_out=True)
        assert isinstance(qo, ops.QuadVolume) and torch.equal(qo.to_planar(), convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, relu=True))
    if Cout in (8, 9):
        rec, logit = convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout, records=True)
        assert torch.equal(rec.t.permute(0, 4, 1, 2, 3), got[:, :8])            # (channels in the order they were packed)
        assert (logit is None) == (Cout == 8) and (logit is None or torch.equal(logit, got[:, 8]))
    # every tiling (rows per wave x planes per workgroup; bmv_tuning BMV_CONV_C4S_RW / _TZ) computes the same sums
    from boostmvsnerfs_amd import _lib
    try:
        for rw in (4, 2):
            for tz in (4, 2):
                _lib.set_tuning("BMV_CONV_C4S_RW", rw), _lib.set_tuning("BMV_CONV_C4S_TZ", tz)
                other = convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout)
                e = float((other.double() - want64).abs().max())
                assert e <= max(2.0 * err32, 1e-6 * scale), f"tiling RW={rw} TZ={tz}: max err {e:.3e} (fp32 kernel {err32:.3e})"
    finally:
        _lib.set_tuning("BMV_CONV_C4S_RW", None), _lib.set_tuning("BMV_CONV_C4S_TZ", None)
    if Cout == 8 and not pair or Cout == 8 and Cin == 16 and sp[0] == 4:
        # the paired and the unpaired form of an 8-channel layer are the same sums in another order
        other = convnet.conv_c4s_fwd(qv, *convnet.pack_conv_c4s(w, b, not pair), Cout)
        _close(other, got, tol=2e-6)


@pytest.mark.parametrize("Cin,Cout,D,H,W", [(16, 8, 2, 8, 12), (16, 8, 3, 9, 19), (6, 5, 1, 4, 33), (32, 4, 2, 5, 16),
                                            (16, 8, 4, 128, 160), (16, 8, 32, 32, 40)])
def test_convT_c4(Cin, Cout, D, H, W):
    """The transposed convolution of the regularisers' last up-sampling step on the 4 x 4 x 1 blocks (csrc/conv_c4.hip)
    against torch's conv_transpose3d and the 16-row engine kernel, with the skip add; every tiling."""
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(Cin + Cout + D)
    x = torch.randn(1, Cin, D, H, W, generator=g).to(DEV)
    w = (torch.randn(Cin, Cout, 3, 3, 3, generator=g) / (Cin * 27 / 8) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    want = F.conv_transpose3d(x, w, b, stride=2, padding=1, output_padding=1)
    skip = torch.randn(want.shape, generator=g).to(DEV)
    wp, bp = convnet.pack_convT_c4(w, b)
    for variant in (0, 1, 2):
        _close(convnet.convT_c4_fwd(x, wp, bp, Cout, variant=variant), want, tol=2e-5)
    _close(convnet.convT_c4_fwd(x, wp, bp, Cout, skip=skip), want + skip, tol=2e-5)
    _close(convnet.convT_c4_fwd(x, wp, bp, Cout, skip=skip, relu=True), F.relu(want) + skip, tol=2e-5)
    wp16, bp16 = convnet.pack_convT(w, b)
    _close(convnet.convT_c4_fwd(x, wp, bp, Cout, skip=skip), convnet.convT3d_fwd(x, wp16, bp16, Cout, skip=skip), tol=2e-5)
    if Cout == 8:     # the result as quad records (what the heads' kernel stages with 16-byte loads): the same bits
        from boostmvsnerfs_amd import ops
        for sk in (None, skip):
            qv = convnet.convT_c4_fwd(x, wp, bp, Cout, skip=sk, relu=True, quad_out=True)
            assert isinstance(qv, ops.QuadVolume)
            assert torch.equal(qv.to_planar(), convnet.convT_c4_fwd(x, wp, bp, Cout, skip=sk, relu=True))
        skq = ops.QuadVolume(skip.view(1, 2, 4, *skip.shape[2:]).permute(0, 1, 3, 4, 5, 2).contiguous())
        assert torch.equal(convnet.convT_c4_fwd(x, wp, bp, Cout, skip=skq, relu=True, quad_out=True).to_planar(),
                           convnet.convT_c4_fwd(x, wp, bp, Cout, skip=skip, relu=True))
        _close(convnet.convT_c4_fwd(x, wp, bp, Cout, skip=skq), want + skip, tol=2e-5)       # (planar out: the skip is converted)


@pytest.mark.parametrize("Cin,Cout,D,H,W", [(8, 16, 8, 64, 80), (8, 16, 8, 256, 320), (8, 16, 64, 64, 80), (4, 16, 5, 9, 19), (12, 9, 3, 17, 33)])
def test_stride2_layer_stages_quad_records(Cin, Cout, D, H, W):
    """The regularisers' stride-2 layer (conv1: 3x3x3, 8 -> 16) behind a first layer that wrote quad records
    (bmv_conv_fwd, out_channels_last | 16): bit-identical to the planar input, every tiling the dispatch picks."""
    from boostmvsnerfs_amd import convnet, ops
    g = torch.Generator().manual_seed(Cin + Cout + D)
    x = torch.randn(1, Cin, D, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    wp, bp = convnet.pack_conv(w, b, 2)
    qv = ops.QuadVolume(x.view(1, Cin // 4, 4, D, H, W).permute(0, 1, 3, 4, 5, 2).contiguous())
    want = convnet.conv_fwd(x, wp, bp, Cout, 3, 3, 2, relu=True)
    _close(want, F.relu(F.conv3d(x, w, b, 2, 1)), tol=2e-5)
    assert torch.equal(convnet.conv_fwd(qv, wp, bp, Cout, 3, 3, 2, relu=True), want)
