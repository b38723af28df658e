import sys, os, json, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from conftest import load_fixture
import test_gpu_mvs as T
from oracle import mvsnerf as M
fx, bfx = load_fixture("mvsnerf_tiny"), load_fixture("boost_mvsnerf_tiny")
def ratios(net, want):
    gmax = max(float(g.abs().max()) for g in want.values())
    out = {}
    for k, p in net.named_parameters():
        err = (p.grad.cpu() - want[k]).abs()
        tol = 2e-3 * want[k].abs() + 2e-3 * float(want[k].pow(2).mean().sqrt()) + 2e-6 * gmax
        out[k] = (float((err / tol).max()), float((err > tol).float().mean()))
    return out
# mvsnerf
cfg = T._cfg(fx, "mvsnerf_eval")
from boostmvsnerfs_amd.networks.mvsnerf.network import Network
sd = fx.group("sd"); batch = fx.batch()
target = torch.rand(1, batch["rays_0"].shape[1], 3, generator=torch.Generator().manual_seed(5))
loss_c, want = T._oracle_grads(lambda s, b: M.mvsnerf_forward(s, b, cfg), sd, batch, target)
for rep in range(3):
    net = T._net(fx, Network)
    bg = {k: (v.to("cuda") if torch.is_tensor(v) else v) for k, v in fx.batch().items()}
    out = net(bg); loss = ((out["rgb_level0"] - target.cuda()) ** 2).mean(); loss.backward()
    r = ratios(net, want)
    top = sorted(r.items(), key=lambda kv: -kv[1][0])[:3]
    print("mvsnerf rep", rep, [(k, round(v[0],3)) for k, v in top])
# boost
tmp = tempfile.mkdtemp()
cfg = T._cfg(bfx, "mvsnerf_ours_eval", tmp)
from boostmvsnerfs_amd.networks.boost_mvsnerf.network import Network as BN
k_best = [int(k) for k in bfx.raw["extra/k_best"]]
json.dump({"synthetic_0": k_best}, open(os.path.join(tmp, "view_selection.json"), "w"))
batch = bfx.batch()
target = torch.rand(1, batch["rays_0"].shape[1], 3, generator=torch.Generator().manual_seed(6))
loss_c, want = T._oracle_grads(lambda s, b: M.boost_mvsnerf_forward(s, b, cfg, k_best), sd, batch, target)
for rep in range(3):
    net = T._net(fx, BN)
    bg = {k: (v.to("cuda") if torch.is_tensor(v) else v) for k, v in bfx.batch().items()}
    out = net(bg); loss = ((out["rgb_level0"] - target.cuda()) ** 2).mean(); loss.backward()
    r = ratios(net, want)
    top = sorted(r.items(), key=lambda kv: -kv[1][0])[:3]
    print("boost rep", rep, [(k, round(v[0],3), round(v[1],5)) for k, v in top])
