import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir("/root/repo/tests")
from conftest import load_fixture, tiny_cfg
from boostmvsnerfs_amd.config import set_cfg
from boostmvsnerfs_amd.networks.enerf.network import Network
from boostmvsnerfs_amd.train import NetworkWrapper
fx = load_fixture("enerf_tiny"); gfx = load_fixture("enerf_tiny_grads")
set_cfg(tiny_cfg(fx, "enerf_pretrain"))
net = Network(); net.load_state_dict(fx.group("sd"), strict=True); net = net.to("cuda").eval()
bg = {k: (v.to("cuda") if torch.is_tensor(v) else v) for k, v in fx.batch().items()}
for i in range(2): bg[f"rgb_{i}"] = gfx.t(f"in/rgb_{i}", "cuda")
_, loss, _, _ = NetworkWrapper(net)(bg); loss.backward()
ref = {k[5:]: torch.from_numpy(v) for k, v in gfx.raw.items() if k.startswith("grad/")}
gmax = max(float(g.abs().max()) for g in ref.values())
rows = []
for k, p in net.named_parameters():
    w = ref[k]; err = (p.grad.cpu() - w).abs(); rms = float(w.pow(2).mean().sqrt())
    for r in (2e-3, 5e-3, 1e-2):
        pass
    f = lambda r: float((err > r * w.abs() + r * rms + 1e-5 * gmax * (r / 1e-2)).float().mean())
    rows.append((f(2e-3), f(5e-3), f(1e-2), float(err.max()) / (rms + 1e-30), rms / gmax, k, w.numel()))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"{r[5]:45s} n={r[6]:6d} out@2e-3 {r[0]:.4f} @5e-3 {r[1]:.4f} @1e-2 {r[2]:.4f} maxerr/rms {r[3]:.2e} rms/gmax {r[4]:.1e}")
print("tensors with any outlier @2e-3:", sum(1 for r in rows if r[0] > 0), "of", len(rows))
