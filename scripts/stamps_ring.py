"""In-kernel step stamps of the ring sweep (build with BMV_RING_DEFS=-DBMV_RING_STAMPS; flags 64 + 4): where a
persistent workgroup's lifetime goes, on the sweep inputs of the headline frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from boostmvsnerfs_amd import ops
from boostmvsnerfs_amd.config import make_cfg, set_cfg
from boostmvsnerfs_amd.synthetic import make_batch
set_cfg(make_cfg("enerf_eval"))
from boostmvsnerfs_amd.networks.enerf.network import Network
dev = "cuda"
torch.manual_seed(0)
net = Network().eval().to(dev)
batch = make_batch(512, 640, device=dev)
calls = []
ops.sweep_hook = lambda impl, args, kw: (calls.append(tuple(t.clone() if torch.is_tensor(t) else t for t in args)), None)[1]
with torch.no_grad():
    net(batch)
ops.sweep_hook = None
extra = int(os.environ.get("STAMP_EXTRA_FLAGS", "0"))
for lvl, (feats, proj, dv) in enumerate(calls):
    variant = int(sys.argv[1 + lvl]) if len(sys.argv) > 1 + lvl else (2 if lvl == 0 else 0)
    cl = feats.permute(0, 1, 3, 4, 2)
    out = torch.zeros(1, cl.shape[-1], *dv.shape[1:], device=dev)
    os.environ["BMV_SWEEP_RING_FLAGS"] = str(64 + 4 + extra)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(300):      # sustained load first: the stamped launch is the last of a back-to-back series
        ops._sweep_variance(cl, proj, dv, algo=100 + variant, channels_last=True, out=out)
    ev[0].record()
    for _ in range(20):
        ops._sweep_variance(cl, proj, dv, algo=100 + variant, channels_last=True, out=out)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"   (eager back-to-back launches with the stamp flags: {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f} us each)")
    del os.environ["BMV_SWEEP_RING_FLAGS"]
    o = out.flatten().cpu().numpy().astype(np.float64)
    o = o[: (o.size // 80) * 80].reshape(-1, 80)
    o = o[(o[:, 3] > 0) & (o[:, 3] < 1000) & (o[:, 2] > 0) & (o[:, 2] < 1e7)]
    n = len(o)
    t0 = o[:, 0] + o[:, 1] * (1 << 24)
    t0 -= t0.min()
    nu = int(np.median(o[:, 3]))
    print(f"== level {lvl} variant {variant}: {n} workgroups, {nu} units each; start spread {t0.max():.0f} cyc; "
          f"lifetime median {np.median(o[:, 2]):.0f}; end max {(t0 + o[:, 2]).max():.0f} cyc; prologue {np.median(o[:, 4]):.0f}")
    cu = o[:, 76] * 1000 + o[:, 78] * 100 + o[:, 79] * 50 + o[:, 77]
    ids, cnt = np.unique(cu, return_counts=True)
    print(f"   distinct CUs {len(ids)}; workgroups per CU histogram {np.bincount(cnt)}; start percentiles (cyc) "
          f"{[int(np.percentile(t0, q)) for q in (0, 25, 50, 75, 90, 100)]}; end percentiles {[int(np.percentile(t0 + o[:, 2], q)) for q in (0, 50, 100)]}")
    gaps = []
    for c in ids:
        tt = np.sort(t0[cu == c])
        gaps += list(np.diff(tt))
    print(f"   start-time gaps between the workgroups of one CU (cycles): {[int(np.percentile(gaps, q)) for q in (0, 25, 50, 75, 100)]}")
    print(f"   lifetime by the 100 MHz counter: median {np.median(o[:, 75]) * 10:.0f} ns -> shader clock {np.median(o[:, 2]) / (np.median(o[:, 75]) * 10) :.2f} GHz")
    sel = o[o[:, 3] == nu]
    names = ["c:begin", "c:blended", "p:issued", "p:landed"]
    tot = {k: 0.0 for k in names}
    # all relative to the consumer's begin stamp of the step
    for st in range(min(3 * nu, 16)):
        c0, c1, p2, p3 = (sel[:, 5 + 4 * st + j] for j in range(4))
        nxt = sel[:, 5 + 4 * (st + 1)] if st + 1 < 3 * nu and 5 + 4 * (st + 1) + 3 < 70 else None
        line = f"   step {st:2d}: blend {np.median(c1 - c0):6.0f}  producer issued at {np.median(p2 - c0):6.0f} landed at {np.median(p3 - c0):6.0f}"
        if nxt is not None:
            line += f"  step length {np.median(nxt - c0):6.0f}"
            tot["c:begin"] += np.median(nxt - c0)
        tot["c:blended"] += np.median(c1 - c0)
        print(line)
    print("   totals per phase (median cycles over the workgroup's life):", {k: int(v) for k, v in tot.items()})
