"""MLP backward (MFMA data path + GEMM weight gradients) vs torch.autograd on the CPU oracle."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _views(img, S):
    """The fixture's (1,P,3,F+4) per-view inputs as S in {2, 3, 4} views (a fourth view = a perturbed copy of the first)."""
    if S <= 3:
        return img[:, :, :S].contiguous()
    g = torch.Generator().manual_seed(5)
    extra = img[:, :, :1] * 0.9 + 0.05 * torch.randn(img[:, :, :1].shape, generator=g)
    return torch.cat([img, extra], 2).contiguous()


@pytest.mark.parametrize("S", [3, 2, 4])
@pytest.mark.parametrize("lvl,feat_ch", [(1, 8), (0, 32)])
def test_nerf_mlp_backward(enerf_fx, lvl, feat_ch, S):
    """Forward and every gradient of the MLP kernels for S = 2, 3, 4 source views (the reference's Agg / NeRF take any
    S, lib/networks/enerf/nerf.py:29-43, 74-89; pre-training draws 2..4, dtu_pretrain.yaml:22-23) vs the oracle."""
    from boostmvsnerfs_amd import autograd as A, ops
    from oracle import enerf as O
    sd = enerf_fx.group("sd")
    prefix = f"nerf_{lvl}."
    names = [f"{prefix}{n}.{k}" for n in ops.NERF_PARAM_ORDER for k in ("weight", "bias")]
    P = 1003                                            # ragged: not a multiple of the 32-sample tile
    vox = enerf_fx.t(f"cap/get_vox_feat#{lvl}")[:, :P].contiguous()
    img = _views(enerf_fx.t(f"cap/get_img_feat#{lvl}")[:, :P], S)
    torch.manual_seed(4)
    g = torch.randn(1, P, 4)
    # CPU oracle + autograd
    sd_cpu = {k: v.detach().clone().requires_grad_(k in names) for k, v in sd.items()}
    v_c, i_c = vox.clone().requires_grad_(True), img.clone().requires_grad_(True)
    out_c = O.nerf_mlp(sd_cpu, prefix, v_c, i_c)
    (out_c * g).sum().backward()
    # HIP
    params = [sd[n].to(DEV).clone().requires_grad_(True) for n in names]
    v_g, i_g = vox.to(DEV).requires_grad_(True), img.to(DEV).requires_grad_(True)
    out_g = A.NerfMLP.apply(v_g, i_g, feat_ch, *params)
    assert_close(out_g, out_c, name="forward")
    with torch.no_grad():       # the inference entry point (bmv_nerf_mlp_fwd) with the cached blob
        blob = ops.nerf_pack_weights([p.detach() for p in params], feat_ch)
        assert torch.equal(ops.nerf_mlp(v_g.detach(), i_g.detach(), blob, feat_ch), out_g.detach())
    (out_g * g.to(DEV)).sum().backward()
    assert_close(v_g.grad, v_c.grad, rtol=2e-3, atol_scale=2e-3, name="d_vox_feat")
    assert_close(i_g.grad, i_c.grad, rtol=2e-3, atol_scale=2e-3, name="d_img_feat")
    # agg_w_fc.bias only shifts the logits of a softmax: its gradient is exactly 0 in real arithmetic and pure
    # rounding noise in fp32, so the absolute floor is tied to the scale of the whole parameter gradient
    gmax = max(float(sd_cpu[n].grad.abs().max()) for n in names)
    for n, p in zip(names, params):
        want = sd_cpu[n].grad
        err = (p.grad.cpu() - want).abs()
        tol = 2e-3 * want.abs() + 2e-3 * float(want.pow(2).mean().sqrt()) + 1e-6 * gmax
        assert bool((err <= tol).all()), f"grad {n}: worst {float(err.max()):.3e} (scale {gmax:.3e})"


def test_nerf_mlp_backward_many_tiles(enerf_fx):
    """70 001 samples (2188 tiles): every workgroup of the data-path kernel and of the weight-gradient kernel walks
    its grid-stride loop several times, the last tile is ragged and the tile count is even + the pair count odd-ish;
    gradients vs torch.autograd on the CPU oracle."""
    from boostmvsnerfs_amd import autograd as A, ops
    from oracle import enerf as O
    sd = enerf_fx.group("sd")
    lvl, feat_ch, prefix = 1, 8, "nerf_1."
    names = [f"{prefix}{n}.{k}" for n in ops.NERF_PARAM_ORDER for k in ("weight", "bias")]
    P = 70001
    torch.manual_seed(11)
    base_v, base_i = enerf_fx.t(f"cap/get_vox_feat#{lvl}"), enerf_fx.t(f"cap/get_img_feat#{lvl}")
    idx = torch.randint(0, base_v.shape[1], (P,))
    vox = (base_v[:, idx] + 0.05 * torch.randn(1, P, 8)).contiguous()
    img = (base_i[:, idx] + 0.05 * torch.randn(1, P, *base_i.shape[2:])).contiguous()
    g = torch.randn(1, P, 4)
    sd_cpu = {k: v.detach().clone().requires_grad_(k in names) for k, v in sd.items()}
    v_c, i_c = vox.clone().requires_grad_(True), img.clone().requires_grad_(True)
    (O.nerf_mlp(sd_cpu, prefix, v_c, i_c) * g).sum().backward()
    params = [sd[n].to(DEV).clone().requires_grad_(True) for n in names]
    v_g, i_g = vox.to(DEV).requires_grad_(True), img.to(DEV).requires_grad_(True)
    (A.NerfMLP.apply(v_g, i_g, feat_ch, *params) * g.to(DEV)).sum().backward()
    assert_close(v_g.grad, v_c.grad, rtol=2e-3, atol_scale=2e-3, name="d_vox_feat")
    assert_close(i_g.grad, i_c.grad, rtol=2e-3, atol_scale=2e-3, name="d_img_feat")
    gmax = max(float(sd_cpu[n].grad.abs().max()) for n in names)
    for n, p in zip(names, params):
        want = sd_cpu[n].grad
        err = (p.grad.cpu() - want).abs()
        tol = 2e-3 * want.abs() + 2e-3 * float(want.pow(2).mean().sqrt()) + 1e-6 * gmax
        assert bool((err <= tol).all()), f"grad {n}: worst {float(err.max()):.3e} (scale {gmax:.3e})"
