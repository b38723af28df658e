"""In-kernel phase stamps of the windowed sweep (flags 64): where a workgroup's lifetime goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from boostmvsnerfs_amd import ops
from boostmvsnerfs_amd.synthetic import make_batch
dev = "cuda"
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
H, W = 512, 640
b = make_batch(H, W, device=dev)
torch.manual_seed(0)
C, D, h, w, Hs, Ws = 16, 8, 256, 320, 256, 320
feats = torch.randn(1, 3, C, Hs, Ws, device=dev)
P = ops.proj_mats(b["src_exts"], b["src_ixts"], b["tar_ext"], b["tar_ixt"], 0.5, 0.5)
dv = (3.0 + 0.5 * torch.rand(1, 1, h, w, device=dev) + torch.linspace(-1.2, 1.2, D, device=dev).view(1, -1, 1, 1)).contiguous()
nhwc = ops.nchw_to_nhwc(feats)
for _ in range(3):
    ops._sweep_variance(nhwc, P, dv, algo=40 + variant, channels_last=True)
os.environ["BMV_SWEEP_WIN_FLAGS"] = "64"
out = torch.zeros(1, C, D, h, w, device=dev)
ops._sweep_variance(nhwc, P, dv, algo=40 + variant, channels_last=True, out=out)
torch.cuda.synchronize()
o = out.flatten().cpu().numpy().astype(np.float64)
n = int((o.reshape(-1, 16)[:, 2] > 0).sum())
o = o[: n * 16].reshape(n, 16)
t0 = o[:, 0] + o[:, 1] * (1 << 24)
t0 -= t0.min()
names = ["dv loaded", "range barrier", "windows", "geometry", "fill0 landed", "blend0", "fill1 landed", "blend1", "fill2 landed", "blend2"]
print(f"variant {variant}: {n} workgroups; start spread {t0.max():.0f} cyc; end max {(t0 + o[:, 11]).max():.0f} cyc")
prev = np.zeros(n)
for i, nm in enumerate(names):
    cur = o[:, i + 2]
    dlt = cur - prev
    print(f"  {nm:14s} +{np.median(dlt):7.0f} (p10 {np.percentile(dlt,10):6.0f} p90 {np.percentile(dlt,90):6.0f})  cum median {np.median(cur):7.0f}")
    prev = cur
starts = np.sort(t0)
print("  start times percentiles (cyc):", [int(np.percentile(t0, q)) for q in (10, 25, 50, 75, 90, 100)])
print("  workgroups per XCC:", np.bincount(o[:, 12].astype(int)))
