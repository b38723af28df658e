import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from conftest import load_fixture
from boostmvsnerfs_amd import autograd as A, ops
from oracle import enerf as O
fx = load_fixture("enerf_tiny")
sd = fx.group("sd")
for lvl, feat_ch, P in ((1, 8, 1003), (1, 8, 12288), (0, 32, 3072)):
    prefix = f"nerf_{lvl}."
    names = [f"{prefix}{n}.{k}" for n in ops.NERF_PARAM_ORDER for k in ("weight", "bias")]
    vox = fx.t(f"cap/get_vox_feat#{lvl}")[:, :P].contiguous()
    img = fx.t(f"cap/get_img_feat#{lvl}")[:, :P].contiguous()
    torch.manual_seed(4)
    g = torch.randn(1, vox.shape[1], 4)
    res = {}
    for dt in (torch.float32, torch.float64):
        sd_c = {k: v.detach().clone().to(dt).requires_grad_(k in names) for k, v in sd.items() if v.is_floating_point()}
        v_c, i_c = vox.clone().to(dt).requires_grad_(True), img.clone().to(dt).requires_grad_(True)
        (O.nerf_mlp(sd_c, prefix, v_c, i_c) * g.to(dt)).sum().backward()
        res[dt] = {n: sd_c[n].grad.double() for n in names}
    params = [sd[n].cuda().clone().requires_grad_(True) for n in names]
    v_g, i_g = vox.cuda().requires_grad_(True), img.cuda().requires_grad_(True)
    (A.NerfMLP.apply(v_g, i_g, feat_ch, *params) * g.cuda()).sum().backward()
    gmax = max(float(res[torch.float64][n].abs().max()) for n in names)
    print(f"level {lvl} P {vox.shape[1]}")
    for n, p in zip(names, params):
        want = res[torch.float64][n]
        tol = 2e-3 * want.abs() + 2e-3 * float(want.pow(2).mean().sqrt()) + 1e-6 * gmax
        r_gpu = float(((p.grad.cpu().double() - want).abs() / tol).max())
        r_cpu32 = float(((res[torch.float32][n] - want).abs() / tol).max())
        print(f"   {n:28s} hip vs f64 {r_gpu:.4f}   oracle f32 vs f64 {r_cpu32:.4f}")
