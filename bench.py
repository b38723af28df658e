#!/usr/bin/env python3
"""Benchmark of the BoostMVSNeRFs rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): rendered Mray/s (512x640 target, 3 source views, 64 depth
planes) = rays rendered by the whole job per second of `Network.forward(batch)`,
bracketed the way the reference's evaluate loop brackets it (run.py:113-129:
synchronize -> t0 -> network(batch) -> synchronize, one step at a time).  A step is
one forward over one synthetic batch already in HBM (other device tensors every step,
as run.py hands them over; `value_extra.resident_batch` is the declared-resident
opt-in that copies nothing); EVERY timed step ends
with a device synchronize, so `value` is the run.py number, not a pipelined one.
The forward that is timed is the drop-in call itself, `net(batch)`, which replays
the HIP graph it captured of its own frame (autograph.AutoGraph; `config.launch`).
`value_extra`: `sync_bracketed_eager` (the same bracket around the ~42 eager
launches), `pipelined_replay` (no per-step synchronize), `value_cold` (the bracket
started after 50 ms of idle, no spin-up), the PCIe-inclusive legs
(`host_batch_sync*`: the batch copied from pinned host memory every frame -- eager,
with the rays built on the device, and through `net(batch)` with new device tensors
per frame as an unchanged run.py loop hands them over), and
`renderer_fp32_mfma` / `all_fp32_mfma` (the same bracket with the frame's bf16 x 3
kernels switched back to fp32 matrix instructions: bmv_tuning BMV_RENDER_SPLIT=0, and
that plus the regularisers' first layers / heads on the fp32 4-row blocks -- `value`
runs the defaults, bf16 MFMAs on three-piece fp32 operands at fp32 accuracy -- each
with that frame's own distance to the oracle).  `parity_max_rel`: the frame the timed steps render against the oracle's
frame of the `cpu_baseline` leg.

N > 1 (one process per GPU, torch.distributed over RCCL): independent target
views are sharded across ranks -- every rank renders its own target frame of the
same scene -- and the rendered (rgb, depth) tiles are all-gathered every step
(the path's only exchange).  Per-GPU work is fixed: weak scaling.
`--shard rays` instead splits ONE frame's rays across ranks (strong scaling of
the renderer only; the front end is replicated).

One JSON line on rank 0, with `roofline` (plane-sweep kernel, HIP-event timed in
the timed region, algorithmic bytes of SURVEY.md section 8d) and `cpu_baseline`
(the CPU oracle -- the port of the reference's PyTorch-CPU path -- on this
host's cores, one frame of the same workload).

Other workloads (`--workload`) time the remaining BASELINE configs; they are not
the headline line and print the same JSON shape without `cpu_baseline` unless
`--cpu-baseline` is given.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)
BF16_MFMA_PEAK_TFLOPS = 2516.6 # v_mfma_f32_32x32x16_bf16 dense peak: 32 768 FLOP / 32 cycles x 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PFLOP/s)

HEADLINE = "enerf_512x640_3src_64planes"
WORKLOADS = {
    # name: dict(net, preset, H, W, planes | samples, views, k_best)
    HEADLINE: dict(net="enerf", preset="enerf_eval", H=512, W=640, planes=[64, 8], views=3),          # configs[1]
    "enerf_256x320_3src_32planes": dict(net="enerf", preset="enerf_eval", H=256, W=320, planes=[32, 8], views=3),
    # the reference's other view counts (train_input_views [2, 3, 4], configs/exps/pretrain/enerf/dtu_pretrain.yaml:22-23)
    "enerf_512x640_2src_64planes": dict(net="enerf", preset="enerf_eval", H=512, W=640, planes=[64, 8], views=2),
    "enerf_512x640_4src_64planes": dict(net="enerf", preset="enerf_eval", H=512, W=640, planes=[64, 8], views=4),
    "enerf_ours_480x736_6src_k4": dict(net="boost_enerf", preset="enerf_ours_eval", H=480, W=736, planes=[64, 8],
                                       views=6, k_best=4),                                           # configs[2]
    "mvsnerf_224x352_32planes": dict(net="mvsnerf", preset="mvsnerf_eval", H=224, W=352, samples=32, views=3),
    "mvsnerf_ours_224x352_128planes_k4": dict(net="boost_mvsnerf", preset="mvsnerf_ours_eval", H=224, W=352,
                                              samples=128, views=6, k_best=4),                       # configs[3]
    # configs[4]: per-scene fine-tune step (forward + backward + Adam), both levels rendered, DDP over RCCL for N>1
    "enerf_ours_ft_480x736_6src_k4": dict(net="boost_enerf", preset="enerf_ours_ft", H=480, W=736, planes=[64, 8],
                                          views=6, k_best=4, train=True),
    "enerf_ft_512x640_3src": dict(net="enerf", preset="enerf_pretrain", H=512, W=640, planes=[64, 8], views=3, train=True),
}


_SELECTION = None   # triplet indices of the K cost volumes (boost workloads)


def sweep_bytes(S, C, Hs, Ws, D, h, w):
    """read every source feature map once + write the variance volume once."""
    return 4 * (S * C * Hs * Ws + C * D * h * w)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps; default 400 for the ENeRF inference workloads (1-4 ms frames: `value` is the MEAN "
                         "of the per-step brackets and a single host hiccup of 2 ms moves a 30-step mean by 7 %%), 30 otherwise")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="views", choices=["views", "rays", "volumes"],
                    help="views: one target frame per rank (weak scaling); rays: ray ranges of ONE frame, front end replicated; "
                         "volumes (K-volume boost networks): cost volumes x ray ranges of ONE frame (sharding.VolumeShard)")
    ap.add_argument("--sweep-algo", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true", help="also time the CPU oracle for non-headline workloads")
    ap.add_argument("--grad-f64", type=int, default=1, help="fine-tune workloads with a cpu_baseline: also run the oracle's step in float64 (parity_max_rel.grad_vs_float64)")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the frame as HIP graphs (inference workloads)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N>1, views sharding: wait for each frame's all-gather before rendering the next frame")
    ap.add_argument("--pipelined", action="store_true",
                    help="do not synchronize after every timed step (the un-bracketed replay rate; NOT the run.py metric)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--event-every", type=int, default=10,
                    help="graph replay: the frame with event brackets is replayed on every Nth timed step, the same "
                         "frame captured without them on the others (six event records cost a 1 ms frame ~3 %%); 1 = "
                         "brackets on every step")
    ap.add_argument("--spinup-steps", type=int, default=100,
                    help="ENeRF inference workloads (ms-scale frames): untimed replays of the step right before the timed "
                         "region, so that it starts from the steady state of a serving renderer (0 = cold start)")
    ap.add_argument("--in-graph-sweeps", action="store_true",
                    help="bracketed frames: keep the frame ONE graph and bracket the plane sweeps with event-record nodes "
                         "instead of launching them between graphs with events bound to their dispatch")
    ap.add_argument("--all-kernel-events", action="store_true", help="HIP events around every hot-path launch")
    ap.add_argument("--miopen-find", type=int, default=0,
                    help="1: let MIOpen search its convolution solvers (torch.backends.cudnn.benchmark).  Only the "
                         "training workloads still run convolutions on MIOpen (inference uses csrc/conv.hip), and "
                         "its search for the 3-D fp32 backward solvers takes > 15 min: off by default")
    ap.add_argument("--stub-renderer", action="store_true",
                    help="TEST ONLY (tests/test_bench_cli.py): CPU tensors, gloo, and a stub in place of the network, so that "
                         "the launcher, the rendezvous, the exchange and the timing contract of `--gpus N` run without a GPU")
    args = ap.parse_args()
    if args.steps is None:
        w = WORKLOADS[args.workload]
        args.steps = 400 if (w["net"] in ("enerf", "boost_enerf") and not w.get("train")) else 30
    return args


def build(args, rank, dev):
    """Network, resident batch and bookkeeping for the chosen workload."""
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    wl = WORKLOADS[args.workload]
    cfg = make_cfg(wl["preset"])
    cc = cfg.enerf.cas_config
    if "planes" in wl:
        cc.volume_planes = list(wl["planes"])
    if "samples" in wl:
        cc.num_samples = [wl["samples"]]
    if "k_best" in wl:
        cc.k_best = wl["k_best"]
    cfg.result_dir = tempfile.mkdtemp(prefix="bmv_bench_")
    set_cfg(cfg)
    H, W = wl["H"], wl["W"]
    mvs = wl["net"] in ("mvsnerf", "boost_mvsnerf")
    tar_offset = (0.05 * rank, 0.0, 0.0) if args.shard == "views" else (0.0, 0.0, 0.0)
    batch_cpu = make_batch(H, W, n_views=wl["views"], seed=0, tar_offset=tar_offset, depth_ranges=mvs,
                           render_scales=(1.0,) if mvs else (0.25, 1.0))
    if mvs:   # a real depth interval in the near/far columns (the shipped loaders put pixel x, y there: quirk 9)
        batch_cpu["rays_0"][..., 6], batch_cpu["rays_0"][..., 7] = 2.2, 7.5
    torch.manual_seed(0)
    if args.stub_renderer:
        if wl["net"] != "enerf" or wl.get("train"):
            raise SystemExit("--stub-renderer drives the ENeRF inference workloads only")
        return cfg, wl, _StubRenderer(1), {}, batch_cpu, clone_batch(batch_cpu, dev), 1
    if wl["net"] == "enerf":
        from boostmvsnerfs_amd.networks.enerf.network import Network
        net = Network()
    elif wl["net"] == "mvsnerf":
        from boostmvsnerfs_amd.networks.mvsnerf.network import Network
        net = Network()
    else:
        if wl["net"] == "boost_enerf":
            from boostmvsnerfs_amd.networks.boost_enerf.network import Network
        else:
            from boostmvsnerfs_amd.networks.boost_mvsnerf.network import Network
        pre = Network(preprocess=True).eval().to(dev)           # offline view selection (untimed, run.py:39-85)
        global _SELECTION
        with torch.no_grad():
            sel = pre.forward_view_selection(clone_batch(batch_cpu, dev))
        key = next(iter(sel))
        if len(sel[key]) < cc.k_best:                            # degenerate synthetic geometry: pad the cover
            sel[key] = (sel[key] + [i for i in range(20) if i not in sel[key]])[: cc.k_best]
        _SELECTION = sel[key]
        with open(os.path.join(cfg.result_dir, "view_selection.json"), "w") as f:
            json.dump(sel, f)
        torch.manual_seed(0)
        net = Network()
    net = net.eval()
    sd_cpu = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    if hasattr(net, "sweep_algo"):
        net.sweep_algo = args.sweep_algo
    level = 0 if mvs else 1
    return cfg, wl, net, sd_cpu, batch_cpu, clone_batch(batch_cpu, dev), level


def cpu_baseline(args, cfg, wl, state_dict, batch_cpu, sel=None):
    """The oracle (CPU port of the reference path) on this host's cores.  ENeRF / MVSNeRF: one full frame.  The
    K-volume workloads: a bounded ray sample -- the front end (features, K cost volumes, regularisers) is run in full,
    the per-ray part on two strided ray subsets, and the frame time is the linear extrapolation t(N) of t(n)."""
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    H, W = wl["H"], wl["W"]
    N = H * W
    threads = torch.get_num_threads()

    def timed(fn):
        t0 = time.perf_counter()
        with torch.no_grad():
            fn()
        return time.perf_counter() - t0

    if wl.get("train"):
        # fine-tune step on the CPU port: forward + MSE loss + torch.autograd backward to every parameter (no optimiser:
        # Adam on 115 small tensors is noise next to it).  ENeRF: one full frame; the K-volume network: the front end in
        # full and two strided ray subsets, extrapolated linearly in the ray count like the inference leg.
        from oracle import enerf as O   # checker / baseline only
        cc = cfg.enerf.cas_config
        k_best = list(sel)[: wl.get("k_best", 1)] if sel is not None else None
        arbitrate = bool(getattr(args, "grad_f64", 1))
        cpu_baseline.last_step64 = cpu_baseline.last_step_small = None

        def step(stride, f64=False):
            b = clone_batch(batch_cpu)
            g = torch.Generator().manual_seed(0)
            for i in range(cc.num):
                b[f"rays_{i}"] = b[f"rays_{i}"][:, ::stride].contiguous()
                b[f"rgb_{i}"] = torch.rand(1, b[f"rays_{i}"].shape[1], 3, generator=g)
            if f64:     # the arbiter: the same step in float64 (an fp32-vs-fp32 distance of 0.09 says nothing by itself)
                b = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
            leaves = {k: (v.detach().double() if f64 and v.is_floating_point() else v.detach().clone())
                      .requires_grad_(v.is_floating_point() and "running" not in k) for k, v in state_dict.items()}
            t0 = time.perf_counter()
            if f64:
                torch.set_default_dtype(torch.float64)     # (the oracle builds its grids with the default dtype)
            try:
                out = (O.enerf_forward(leaves, b, cfg) if wl["net"] == "enerf" else O.boost_enerf_forward(leaves, b, cfg, k_best))
                loss = sum(cc.loss_weight[i] * ((out[f"rgb_level{i}"] - b[f"rgb_{i}"]) ** 2).mean()
                           for i in range(cc.num) if f"rgb_level{i}" in out)
                loss.backward()
            finally:
                torch.set_default_dtype(torch.float32)
            dt_ = time.perf_counter() - t0
            rec = {"stride": stride, "loss": float(loss.detach()),
                   "grads": {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None}}
            if f64:
                cpu_baseline.last_step64 = rec
            else:
                # the checker's step: bench compares the HIP step on the same rays / targets with it (parity_max_rel)
                cpu_baseline.last_step = rec
            return dt_, b[f"rays_{cc.num - 1}"].shape[1]
        if wl["net"] == "enerf":
            timed(lambda: O.enerf_forward(state_dict, make_batch(64, 96), cfg))      # page-in / thread-pool warm-up
            dt, _ = step(1)
            sample = f"1 step at {H}x{W} (whole workload: forward + loss + backward), oracle/enerf.py torch-CPU fp32 autograd, {dt:.2f} s"
            if arbitrate:
                step(1, f64=True)
        else:
            (t1, n1) = step(8)
            if arbitrate:        # float64 twin of the SMALLER subset's step; its fp32 step is kept next to it
                cpu_baseline.last_step_small = cpu_baseline.last_step
                step(8, f64=True)
            (t2, n2) = step(4)
            per_ray = max((t2 - t1) / (n2 - n1), 0.0)
            dt = t1 + per_ray * (N - n1)
            sample = (f"forward + loss + backward; front end in full + rays ::8 ({n1} rays, {t1:.2f} s) and ::4 ({n2} rays, "
                      f"{t2:.2f} s), step time extrapolated linearly in the ray count to {N} rays = {dt:.1f} s; oracle torch-CPU fp32 autograd")
    elif wl["net"] == "enerf":
        from oracle import enerf as O   # checker / baseline only
        timed(lambda: O.enerf_forward(state_dict, make_batch(64, 96), cfg))          # page-in / thread-pool warm-up
        keep = {}
        dt = timed(lambda: keep.update(O.enerf_forward(state_dict, clone_batch(batch_cpu), cfg)))
        sample = f"1 frame {H}x{W} (whole workload), oracle/enerf.py torch-CPU fp32, {dt:.2f} s"
        cpu_baseline.last_frame = keep           # the checker's frame: bench compares the GPU frame with it (parity_max_rel)
    elif wl["net"] == "mvsnerf" and wl["samples"] <= 32:
        from oracle import mvsnerf as M
        dt = timed(lambda: M.mvsnerf_forward(state_dict, clone_batch(batch_cpu), cfg))
        sample = f"1 frame {H}x{W} x {wl['samples']} samples (whole workload), oracle/mvsnerf.py torch-CPU fp32, {dt:.2f} s"
    else:
        k_best = list(sel)[: wl.get("k_best", 1)] if sel is not None else None

        def run(stride):
            b = clone_batch(batch_cpu)
            keep = {}
            if wl["net"] in ("mvsnerf", "boost_mvsnerf"):
                from oracle import mvsnerf as M
                b["rays_0"] = b["rays_0"][:, ::stride].contiguous()
                fn = ((lambda: keep.update(M.mvsnerf_forward(state_dict, b, cfg))) if wl["net"] == "mvsnerf" else
                      (lambda: keep.update(M.boost_mvsnerf_forward(state_dict, b, cfg, k_best))))
                n = b["rays_0"].shape[1]
            else:
                from oracle import enerf as O
                for i in range(cfg.enerf.cas_config.num):
                    b[f"rays_{i}"] = b[f"rays_{i}"][:, ::stride].contiguous()
                fn = lambda: keep.update(O.boost_enerf_forward(state_dict, b, cfg, k_best))     # noqa: E731
                n = b[f"rays_{cfg.enerf.cas_config.num - 1}"].shape[1]
            dt_ = timed(fn)
            # the checker's frame on this ray subset: bench renders the same subset on the GPU (parity_max_rel)
            cpu_baseline.last_subset = {"stride": stride, "frame": {k: v for k, v in keep.items() if torch.is_tensor(v)}}
            return dt_, n
        strides = (64, 32) if wl["net"] != "boost_enerf" else (8, 4)
        (t1, n1), (t2, n2) = run(strides[0]), run(strides[1])
        per_ray = max((t2 - t1) / (n2 - n1), 0.0)
        dt = t1 + per_ray * (N - n1)
        sample = (f"front end in full + rays ::{strides[0]} ({n1} rays, {t1:.2f} s) and ::{strides[1]} ({n2} rays, {t2:.2f} s), "
                  f"frame time extrapolated linearly in the ray count to {N} rays = {dt:.1f} s; oracle torch-CPU fp32")
    return {"value": N / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port", "sample": sample,
            "host_cpus": os.cpu_count()}



def _rel_err(got, want):
    """max |d| / (|want| + rms(want)) and how many entries sit outside the 1e-3 bar (a K-volume frame's visibility
    test is a step function: a sample within an ulp of a viewport edge may fall on the other side and move its ray by
    O(1/K); the tests count such flips, tests/test_gpu_fullsize.py: 0 measured)."""
    g = got.detach().float().cpu().reshape(want.shape)
    rms = float(want.pow(2).mean().sqrt())
    rel = (g - want).abs() / (want.abs() + rms + 1e-30)
    return float(rel.max()), int((rel > 1e-3).sum()), rel.numel()


def parity_objects(cfg, wl, net, sd_cpu, batch, batch_cpu, split_frame, dev):
    """`parity_max_rel` of the line: what the timed network renders (or, for the fine-tune workloads, the loss and the
    parameter gradients of its step) against the oracle results the cpu_baseline leg just computed, on the same weights
    and inputs.  ENeRF / MVSNeRF-32: the whole frame; the K-volume workloads: the larger of the two strided ray subsets
    the oracle rendered (the front end is run in full by both sides); training: the larger ray subset's step."""
    from boostmvsnerfs_amd.synthetic import clone_batch
    out = {}
    cc = cfg.enerf.cas_config
    ref = getattr(cpu_baseline, "last_frame", None)
    sub = getattr(cpu_baseline, "last_subset", None)
    stp = getattr(cpu_baseline, "last_step", None)
    if wl.get("train") and stp is not None:
        from boostmvsnerfs_amd.train import NetworkWrapper
        was = net.training
        net.load_state_dict(sd_cpu)                 # the timed steps trained: back to the weights the oracle's step used
        net.eval()                                  # the oracle's step runs the batch norms in eval mode

        def hip_step(stride):
            b = clone_batch(batch_cpu)
            g = torch.Generator().manual_seed(0)
            for i in range(cc.num):
                b[f"rays_{i}"] = b[f"rays_{i}"][:, ::stride].contiguous()
                b[f"rgb_{i}"] = torch.rand(1, b[f"rays_{i}"].shape[1], 3, generator=g)
            b = clone_batch(b, dev)
            net.zero_grad(set_to_none=True)
            _, loss_, _, _ = NetworkWrapper(net)(b)
            loss_.mean().backward()
            grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
            net.zero_grad(set_to_none=True)
            return float(loss_.detach().mean()), grads

        loss, hip = hip_step(stp["stride"])
        worst, worst_name, n_tensors, rels = 0.0, None, 0, []
        gmax = max(float(v.abs().max()) for v in stp["grads"].values())
        for k, gk in hip.items():
            want = stp["grads"].get(k)
            if want is None:
                continue
            num = float((gk - want).pow(2).sum().sqrt())
            rel = num / (float(want.pow(2).sum().sqrt()) + 1e-6 * gmax)
            n_tensors += 1
            rels.append(rel)
            if rel > worst:
                worst, worst_name = rel, k
        # float64 arbitration: the same step in float64 on the CPU; HIP and the fp32 oracle each against it
        arb = None
        s64 = getattr(cpu_baseline, "last_step64", None)
        if s64 is not None:
            small = getattr(cpu_baseline, "last_step_small", None) or stp
            hip64 = hip if s64["stride"] == stp["stride"] else hip_step(s64["stride"])[1]
            g64max = max(float(v.abs().max()) for v in s64["grads"].values())
            rows = []
            for k, w64 in s64["grads"].items():
                if k not in hip64 or k not in small["grads"]:
                    continue
                den = float(w64.pow(2).sum().sqrt()) + 1e-6 * g64max
                rows.append((k, float((hip64[k].double() - w64).pow(2).sum().sqrt()) / den,
                             float((small["grads"][k].double() - w64).pow(2).sum().sqrt()) / den,
                             float((hip64[k] - small["grads"][k]).pow(2).sum().sqrt()) / den))
            rows.sort(key=lambda r: -r[3])
            arb = {"stride": s64["stride"], "tensors": len(rows),
                   "hip_vs_f64_max": max(r[1] for r in rows), "oracle32_vs_f64_max": max(r[2] for r in rows),
                   "hip_vs_f64_median": sorted(r[1] for r in rows)[len(rows) // 2],
                   "oracle32_vs_f64_median": sorted(r[2] for r in rows)[len(rows) // 2],
                   "hip_no_farther_than_oracle32_on": sum(r[1] <= r[2] for r in rows),
                   "hip_within_1.5x_of_oracle32_on": sum(r[1] <= 1.5 * r[2] + 1e-7 for r in rows),
                   "worst_five_by_hip_vs_oracle32": [{"tensor": r[0], "hip_vs_f64": r[1], "oracle32_vs_f64": r[2], "hip_vs_oracle32": r[3]}
                                                     for r in rows[:5]],
                   "what": "relative L2 distance of a parameter gradient to the SAME step evaluated in float64 on the CPU (oracle, "
                           f"rays ::{s64['stride']}): the HIP step and the fp32 oracle step side by side.  An fp32-vs-fp32 distance is "
                           "only as meaningful as the fp32 oracle's own distance to float64 on that tensor"}
        net.zero_grad(set_to_none=True)
        net.train(was)
        out["parity_max_rel"] = {
            "loss": abs(float(loss) - stp["loss"]) / abs(stp["loss"]), "grad_rel_l2_max": worst, "grad_worst_tensor": worst_name,
            "grad_tensors": n_tensors, "grad_rel_l2_median": sorted(rels)[len(rels) // 2] if rels else None,
            "grad_tensors_over_1e-2": sum(r > 1e-2 for r in rels),
            "grad_vs_float64": arb,
            "max": max(worst, abs(float(loss) - stp["loss"]) / abs(stp["loss"])),
            "against": (f"oracle forward + MSE loss + torch.autograd backward (eval-mode batch norm) on rays ::{stp['stride']} of the "
                        "workload's frame, same weights and targets: relative loss difference and the worst per-tensor relative "
                        "L2 distance of the parameter gradients.  Both sides are fp32: at these sizes the fp32 oracle is itself "
                        "up to 1e-2 (relative L2) away from its float64 run on the deep levels' batch-norm and convolution "
                        "parameters (sums over 1e5-1e6 voxels that cancel), tests/test_gpu_fullsize.py arbitrates in float64"),
            "tolerance": "tests: 2e-3 per entry (relative + of the tensor's rms), tests/test_gpu_training.py"}
        return out
    if ref and not wl.get("train"):
        # the frame the timed steps rendered against the oracle's frame of the same weights and batch:
        # max |d| / (|want| + rms(want)) per output (the tests' bar is 1e-3 on it)
        with torch.no_grad():
            got = net(batch)
        par = {k: _rel_err(got[k], want)[0] for k, want in ref.items() if k in got and torch.is_tensor(want)}
        out["parity_max_rel"] = {"per_output": par, "max": max(par.values()) if par else None,
                                 "against": "oracle/enerf.py enerf_forward on the same weights and batch (the cpu_baseline frame)",
                                 "tolerance": 1e-3}
        if split_frame is not None:       # the split-bf16 experiment's frame against the same oracle frame
            ps = {k: _rel_err(split_frame[k], want)[0] for k, want in ref.items() if k in split_frame and torch.is_tensor(want)}
            out["parity_max_rel_split"] = {"per_output": ps, "max": max(ps.values()) if ps else None}
    elif sub and not wl.get("train"):
        b = clone_batch(batch_cpu)
        for k in list(b):
            if k.startswith("rays_"):
                b[k] = b[k][:, ::sub["stride"]].contiguous()
        with torch.no_grad():
            got = net(clone_batch(b, dev))
        torch.cuda.synchronize()
        par, outside = {}, {}
        for k, want in sub["frame"].items():
            if k in got and torch.is_tensor(got[k]) and got[k].numel() == want.numel():
                m, bad, n = _rel_err(got[k], want)
                par[k] = m
                outside[k] = f"{bad}/{n}"
        out["parity_max_rel"] = {"per_output": par, "max": max(par.values()) if par else None, "entries_outside_1e-3": outside,
                                 "against": (f"the oracle's frame of rays ::{sub['stride']} (front end in full), same weights, batch and "
                                             "view selection: the cpu_baseline leg's larger ray subset, rendered by the timed network too"),
                                 "tolerance": 1e-3}
    return out


def make_exchange(shard, world, rank, N, dev, net, wl, k_best, level, pipelined=False, sync_gather=False):
    """The exchange step of a sharding mode on one rank: sets the network's ray / volume shard and returns
    (finish, gather, vshard) -- `finish(out)` turns this rank's forward output into the gathered frame(s) with ONE
    collective per step (all-gather of (rgb, depth) tiles; for `volumes` an all-to-all of the K-volume stacks first).
    World 1: identity.  Runs on any torch.distributed backend (RCCL here, gloo in tests/test_sharding_gloo.py, which
    drives this very function with a stub renderer)."""
    from boostmvsnerfs_amd import sharding
    rgb_key, depth_key = f"rgb_level{level}", f"depth_level{level}"
    if shard == "rays" and world > 1:
        net.ray_range = sharding.ray_slice(N, world, rank)
    vshard = None
    if shard == "volumes" and world > 1:
        if wl["net"] != "boost_enerf" or wl.get("train"):
            raise SystemExit("--shard volumes is the cost-volume parallelism of the boost_enerf inference workloads")
        vshard = sharding.VolumeShard(world, rank, k_best, N)
        net.volume_ids, net.ray_range = vshard.volumes, vshard.ray_range
    gather = sharding.TileGather(world, N if shard == "views" else None, dev) if (world > 1 and vshard is None) else None

    def finish(out):
        if vshard is not None:
            raws, zs, ms = vshard.exchange(*out[f"stacks_level{level}"])
            fused = net.merge_mlp_outputs(raws, ms, zs)
            return vshard.gather_tiles(fused["rgb"], fused["depth"])
        if gather is not None:
            if shard == "views":
                if sync_gather or not pipelined:
                    return gather.all_gather_frames(out[rgb_key], out[depth_key])
                # the exchange of frame i runs on RCCL's stream under the kernels of frame i+1
                return gather.all_gather_frames_pipelined(out[rgb_key], out[depth_key])
            return gather.all_gather_ray_tiles(out[rgb_key], out[depth_key], N)
        return out
    return finish, gather, vshard


class _StubRenderer:
    """--stub-renderer: what the timed step touches of a network -- the ray / volume shard attributes and a forward that
    returns (rgb, depth) of the frame's shape computed from the batch's rays -- so that `bench.py --gpus N` runs end to
    end on CPU + gloo (launcher, rendezvous, exchange, barriers, max-over-ranks timing, the JSON line)."""
    ray_range = None
    volume_ids = None
    training = False

    def __init__(self, level):
        self.level = level

    def __call__(self, batch):
        rays = batch[f"rays_{self.level}"]
        if self.ray_range is not None:
            rays = rays[:, self.ray_range[0]:self.ray_range[1]]
        return {f"rgb_level{self.level}": rays[..., :3] * 0.5, f"depth_level{self.level}": rays[..., 3].clone()}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves, the way the reference's
    train_net.py expects to be started (train_net.py:144-149 reads RANK / LOCAL_RANK from torch.distributed's launcher).
    Only `import torch` has happened in this process -- nothing has touched the GPU -- and the ranks are CHILD processes
    (never an exec of this one); rank 0's JSON line passes through on stdout, the exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL between the ranks (see the task notes)
    env.setdefault("OMP_NUM_THREADS", "1" if not args.stub_renderer else "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    stub = args.stub_renderer
    import torch.distributed as dist
    if stub:
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a, **k: None       # (this process only: the stub run has no device to wait for)
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        torch.backends.cudnn.benchmark = bool(args.miopen_find)   # training-mode convolutions run on MIOpen
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if stub:
            dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", init_method="env://", rank=rank, world_size=world, device_id=dev)

    from boostmvsnerfs_amd import ktimer, sharding
    if stub:
        args.no_kernel_events, args.no_cpu_baseline, args.spinup_steps = True, True, min(args.spinup_steps, 2)

    cfg, wl, net, sd_cpu, batch_cpu, batch, level = build(args, rank, dev)
    cc = cfg.enerf.cas_config
    H, W = wl["H"], wl["W"]
    N = H * W
    rgb_key, depth_key = f"rgb_level{level}", f"depth_level{level}"
    finish, gather, vshard = make_exchange(args.shard, world, rank, N, dev, net, wl, int(cc.k_best) if "k_best" in wl else 1,
                                           level, pipelined=args.pipelined, sync_gather=args.sync_gather)

    if wl.get("train"):
        # fine-tune step (trainer.py:44-63): every rank trains on its own target view, DDP averages the
        # gradients over RCCL (bucketed all-reduce overlapped with backward); no tile gather in training
        from boostmvsnerfs_amd.train import GraphedTrainStep, NetworkWrapper, make_optimizer, train_step
        gen = torch.Generator().manual_seed(rank)
        for i in range(cc.num):
            batch[f"rgb_{i}"] = torch.rand(1, batch[f"rays_{i}"].shape[1], 3, generator=gen).to(dev)
        net.train()
        wrapper = NetworkWrapper(net)
        if world > 1:
            wrapper = torch.nn.parallel.DistributedDataParallel(
                torch.nn.SyncBatchNorm.convert_sync_batchnorm(wrapper), device_ids=[local_rank],
                output_device=local_rank, find_unused_parameters=True)
        optimizer = make_optimizer(net)
        gather = None
        # forward + loss + backward as one HIP graph (clip + Adam eager); DDP's all-reduce hooks are host callbacks, so
        # N > 1 stays eager
        graphed_train = GraphedTrainStep(wrapper, optimizer) if (args.graph and world == 1) else None

        def step():
            if graphed_train is not None:
                return graphed_train(batch)
            return train_step(wrapper, optimizer, batch)
    else:
        # What run.py times (run.py:113-123): every iteration hands `network(batch)` device tensors that are NOT the ones
        # of the previous call (`batch[k] = batch[k].cuda()` in front of the bracket).  The timed steps walk a ring of
        # three distinct device copies of the batch, created here, outside every bracket: `net(...)` sees other
        # tensors on every call, copies them into its graph's private inputs and returns fresh outputs.
        from boostmvsnerfs_amd.synthetic import clone_batch as _clone
        ring = [batch] + [_clone(batch, dev) for _ in range(2)]
        ring_pos = [0]

        def feed():
            ring_pos[0] = (ring_pos[0] + 1) % len(ring)
            return ring[ring_pos[0]]

        def step():
            with torch.no_grad():
                return finish(net(feed()))          # the drop-in call: replays its own HIP graph from the 2nd call on

        def step_plain():
            with torch.no_grad():
                return finish(net(feed()))

        _eager_call = getattr(net, "_forward_checked", net)    # the same frame as ~42 eager launches (BMV_AUTOGRAPH=0)

        def eager_only_step():
            with torch.no_grad():
                return finish(_eager_call(dict(feed())))

    # MIOpen's solver search writes its per-user find-db: rank 0 searches first with a collective-free forward,
    # the other ranks then hit the finished db instead of N processes searching (and locking the db) at once.
    if world > 1 and args.miopen_find and not wl.get("train"):
        def local_forward():
            with torch.no_grad():
                net(batch)
            torch.cuda.synchronize()
        if rank == 0:
            local_forward()
        dist.barrier()
        if rank != 0:
            local_forward()
        dist.barrier()
    # ---- warm-up (eager).  The first step pays module loads / allocator growth: kernel events start after it.
    ktimer.reset()
    ktimer.only = ("render_rays", "mvs_render", "sweep_variance", "mvs_sweep", "empty_bracket")
    for i in range(args.warmup):
        ktimer.enabled = (not args.no_kernel_events) and i > 0
        step()
    torch.cuda.synchronize()
    warm_kernels = ktimer.summary()
    ktimer.enabled = False
    # Inference workloads: replay the frame as HIP graphs (the ~45 launches of a frame cost the host about as long
    # as the GPU needs to run them).  The plane sweeps stay ordinary launches between the graphs so the HIP events
    # of `roofline` time them inside the timed region.  Falls back to eager launches if capture fails.
    graph_note = "off"
    split_frame = None
    side_frames = {}
    if not wl.get("train"):
        graphed_train = None
    eager_step = step
    sampled = {"n": 0, "every": 1, "evented": True}
    if not wl.get("train"):
        eager_step = eager_only_step
        if not (args.graph and hasattr(net, "_autograph")) or args.all_kernel_events:
            step = eager_only_step           # --no-graph / per-launch events: every timed step is eager launches
    if args.graph and not wl.get("train") and not args.all_kernel_events and wl["net"] in ("enerf", "boost_enerf") and hasattr(net, "_autograph"):
        try:
            from boostmvsnerfs_amd.framegraph import FrameGraph
            # the kernels of `roofline` are bracketed by event-record nodes INSIDE the one graph of a frame
            # (csrc/timing.hip); --cut-sweeps restores round 1's form (the sweeps as ordinary launches between graphs)
            ktimer.forget_graph_events()
            sampled = {"n": 0, "every": 1, "evented": True}
            if args.no_kernel_events:
                fg = None
                replay = lambda: net(feed())   # noqa: E731  (Network.forward replays its own graph)
            else:
                # bracketed frames: the sweeps are ordinary launches between the graphs, their events bound to their
                # own dispatch (hipExtLaunchKernelGGL: the kernel's begin and end); the renderer is bracketed by
                # event-record nodes inside its graph.  --in-graph-sweeps keeps the frame one graph and brackets the
                # sweeps with event-record nodes too (they then read ~2.5 us more).
                fg = FrameGraph(net, _clone(batch, dev), cut=None if args.in_graph_sweeps else "all", events=True)
                replay = fg.replay
                if fg.events and args.event_every > 1:
                    # plain steps are the drop-in call itself, `net(batch)`: Network.forward replays the graph it
                    # captured of the same frame (autograph.AutoGraph); every `event_every`-th step replays the
                    # bracketed capture instead (the measurement instrument)
                    sampled["every"] = args.event_every

                    from boostmvsnerfs_amd import autograph as _ag
                    reads = net._autograph_inputs(batch) if hasattr(net, "_autograph_inputs") else None
                    fg_names = [k for k, v in fg.batch.items() if torch.is_tensor(v) and not getattr(v, "_bmv_built_rays", False)
                                and (reads is None or k in reads)]      # (the K-volume forward REPLACES batch['src_*']: quirk 8)

                    def replay():   # noqa: F811
                        sampled["evented"] = sampled["n"] % sampled["every"] == 0
                        sampled["n"] += 1
                        b = feed()
                        if not sampled["evented"]:
                            return net(b)
                        # the bracketed capture does what net(b) does: inputs copied in, frame replayed, outputs copied out
                        if b is not fg.batch:
                            _ag._copy_many([fg.batch[k] for k in fg_names], [b[k] for k in fg_names])
                        return _ag.AutoGraph._fresh_outputs({}, fg.replay())

            def step():   # noqa: F811
                with torch.no_grad():
                    return finish(replay())
            for _ in range(4):
                step()
            ag = net._autograph.stats
            graph_note = (f"net(batch) on other device tensors every step (a ring of {len(ring)}), replaying its own HIP graph (autograph: "
                          f"{ag['captures']} capture(s); in {ag['replays']} replays {ag.get('deferred', 0)} large inputs / outputs read and written "
                          f"in place through the frame's pointer table, {ag['copies']} small input tensors copied)"
                          + ("" if fg is None else f"; every {sampled['every']}. step a bracketed capture instead: {len(fg.graphs)} graph(s)"
                             + (f" + {len(fg.sweeps)} eager plane sweep(s) with dispatch-bound events" if fg.sweeps else "")))
        except Exception as e:   # keep the eager path: the bench must still produce its line
            print(f"[bench] HIP-graph capture failed, staying eager: {type(e).__name__}: {e}", file=sys.stderr)
            graph_note = f"capture failed ({type(e).__name__})"
            step = eager_step
            torch.cuda.synchronize()

    def bracketed(fn, iters):
        """run.py:113-129: synchronize, t0, network(batch), synchronize, per step; the first iteration is dropped."""
        times = []
        for _ in range(iters + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return sum(times[1:]) / iters

    extra = {}
    if not wl.get("train"):
        n_x = max(5, min(args.steps, 20))
        t_eager = bracketed(eager_step, n_x)
        extra["sync_bracketed_eager"] = {"value": N * (world if args.shard == "views" else 1) / t_eager / 1e6, "ms_per_step": t_eager * 1e3,
                                         "what": "run.py:113-129 bracket around the eager net(batch) (what a drop-in run.py calls)"}
        if step is not eager_step:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_x):
                step()
            if gather is not None:
                gather.flush()
            torch.cuda.synchronize()
            t_pipe = (time.perf_counter() - t0) / n_x
            extra["pipelined_replay"] = {"value": N * (world if args.shard == "views" else 1) / t_pipe / 1e6, "ms_per_step": t_pipe * 1e3,
                                         "what": "graph replays issued back to back, one synchronize at the end (not the metric)"}

    if not wl.get("train") and world == 1 and not stub:
        # the reference's loaders hand over HOST tensors (run.py:114-116 moves them every frame): the same bracket with
        # the batch copied from pinned host memory each step, with the rays copied too / built on the device instead
        reads_h = net._autograph_inputs(batch) if hasattr(net, "_autograph_inputs") else None
        host = {k: v.cpu().pin_memory() for k, v in batch.items() if torch.is_tensor(v) and (reads_h is None or k in reads_h)}
        ray_keys = [k for k in host if k.startswith("rays_")]
        hb = _clone(batch, dev)                         # the legs' own working batch (they pop / refill its keys)

        def from_host(device_rays):
            def fn():
                for k, v in host.items():
                    if device_rays and k in ray_keys:
                        hb.pop(k, None)                 # Network.ensure_rays rebuilds them from tar_ext / tar_ixt
                    else:
                        if k not in hb or hb[k].shape != v.shape:      # (the K-volume forwards replace batch['src_*'])
                            hb[k] = torch.empty_like(v, device=dev)
                        hb[k].copy_(v, non_blocking=True)
                with torch.no_grad():
                    return finish(_eager_call(hb))
            return fn
        for name, dr in (("host_batch_sync_eager", False), ("host_batch_device_rays_sync_eager", True)):
            t_h = bracketed(from_host(dr), n_x)
            extra[name] = {"value": N / t_h / 1e6, "ms_per_step": t_h * 1e3,
                           "what": "run.py bracket incl. the host->device copy of the batch (PCIe), eager forward"
                                   + ("; rays built on the device (bmv_make_rays) instead of copied" if dr else "")}
        if args.graph and hasattr(net, "_autograph") and not args.all_kernel_events:
            # ... and what an unchanged run.py loop gets: every frame a NEW set of device tensors (`batch[k].cuda()`,
            # run.py:114-116) handed to net(batch), which copies them into its captured buffers and replays
            meta = {k: v for k, v in batch.items() if not torch.is_tensor(v)}

            def run_py_frame():
                fresh = dict(meta)
                for k, v in host.items():
                    fresh[k] = v.to(dev, non_blocking=True)
                with torch.no_grad():
                    return finish(net(fresh))
            # This batch has other keys than the resident one (only what the frame reads travels), so it is a new key for
            # the network's AutoGraph: frame 0 runs eagerly, frame 1 CAPTURES (tens of ms, charged to that call), frame 2
            # is the first replay.  Rounds 4-5 timed 20 frames from frame 1 on -- the capture inside the bracket was 2.5 of
            # the 3.7 ms "per step" (scripts/probe_host_batch.py prints the series).  Like every other leg
            # the steady state is what is timed: three untimed frames first, the capture's cost reported beside it.
            ag_stats = dict(net._autograph.stats)
            t_first = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_py_frame()
                torch.cuda.synchronize()
                t_first.append((time.perf_counter() - t0) * 1e3)
            t_h = bracketed(run_py_frame, n_x)
            ag_delta = {k: net._autograph.stats.get(k, 0) - ag_stats.get(k, 0) for k in ("eager", "captures", "replays")}
            extra["host_batch_sync"] = {"value": N / t_h / 1e6, "ms_per_step": t_h * 1e3,
                                        "first_frames_ms": [round(t, 3) for t in t_first],
                                        "autograph": ag_delta,
                                        "what": "run.py bracket incl. the host->device copy of the batch (PCIe) into NEW device "
                                                "tensors every frame, net(batch) = pointer-table feed + graph replay (steady "
                                                "state: `first_frames_ms` are the untimed eager / capturing / first replayed "
                                                "frames of this batch structure; `autograph` counts all of the leg's calls)"}
        from boostmvsnerfs_amd import _lib as _bl
        if args.graph and hasattr(net, "_autograph") and not args.all_kernel_events and wl["net"] == "enerf":
            # The same frame with its bf16 x 3 kernels switched back to fp32 matrix instructions.  `value` runs the defaults:
            # the renderer MLP's two-tile chains (csrc/mlp.hpp CSPLIT, round 5) and the regularisers' first layers and
            # heads (csrc/conv_c4s.hip, round 6) as bf16 MFMAs on three-piece fp32 operands -- the fp32 values exactly,
            # product terms of at most 2^-23 of a product dropped, fp32 accumulation; accuracy against float64:
            # profiles/r5/mlp_split_accuracy.txt, tests/test_gpu_conv.py::test_conv_c4s.
            #   renderer_fp32_mfma: bmv_tuning BMV_RENDER_SPLIT=0 (every chain of the MLP on fp32 MFMAs);
            #   all_fp32_mfma: that AND cost_reg_i.conv_c4s = False (the fp32 4-row blocks of csrc/conv_c4.hip): no bf16
            #   matrix instruction in the frame -- what `value` measured in rounds 1-4.
            regs = [getattr(net, f"cost_reg_{i}") for i in range(cc.num) if hasattr(net, f"cost_reg_{i}")]
            c4s_was = [r.conv_c4s for r in regs]
            rs_was = _bl.get_tuning("BMV_RENDER_SPLIT")        # (an environment / caller setting survives)
            legs = []
            if rs_was != 0:
                legs.append(("renderer_fp32_mfma", 0, None,
                             "same bracket with bmv_tuning BMV_RENDER_SPLIT=0: every chain of the renderer's MLP on fp32 MFMAs"))
            if rs_was != 0 or any(c4s_was):
                legs.append(("all_fp32_mfma", 0, False,
                             "same bracket with NO bf16 matrix instruction in the frame: BMV_RENDER_SPLIT=0, the regularisers' "
                             "first layers / heads on the fp32 4-row blocks (cost_reg_i.conv_c4s = False) and FeatureNet on the fp32 "
                             "engine (BMV_CONV0_S = BMV_CONV2D_S = BMV_FPN_S = 0): what `value` measured in rounds 1-4"))
            import contextlib

            from boostmvsnerfs_amd import switches as _sw
            for name, rsplit, c4s, what in legs:
                _bl.set_tuning("BMV_RENDER_SPLIT", rsplit)
                if c4s is not None:
                    for r in regs:
                        r.conv_c4s = c4s
                net._autograph.invalidate()              # (the captured frame has the other renderer baked in)
                # (... and FeatureNet's bf16 x 3 kernels of round 6 -- first block, encoder, last top-down step -- back on fp32 MFMAs)
                fp32_cnn = _sw.override(BMV_CONV0_S=0, BMV_CONV2D_S=0, BMV_FPN_S=0) if c4s is False else contextlib.nullcontext()
                try:
                    with fp32_cnn:
                        for _ in range(4):                  # eager, capture, first replays of the new configuration
                            step_plain()
                        torch.cuda.synchronize()
                        t_r = bracketed(step_plain, max(n_x, 50))
                        with torch.no_grad():
                            side_frames[name] = {k: v.detach().float().cpu() for k, v in net(batch).items() if torch.is_tensor(v)}
                        extra[name] = {"value": N / t_r / 1e6, "ms_per_step": t_r * 1e3,
                                       "parity_max_rel": None,     # (filled in behind the cpu_baseline leg, which renders the oracle's frame)
                                       "what": what + "; parity_max_rel = its frame against the oracle's"}
                except Exception as e:                  # a side measurement must not take the metric's line down with it
                    extra[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                    side_frames.pop(name, None)
            _bl.set_tuning("BMV_RENDER_SPLIT", rs_was)
            for r, w_ in zip(regs, c4s_was):
                r.conv_c4s = w_
            if legs:
                try:                                    # (back to the default configuration: its own failure is recorded,
                    net._autograph.invalidate()         # not raised -- the metric's timed region warms up again anyway)
                    for _ in range(3):
                        step_plain()
                    torch.cuda.synchronize()
                except Exception as e:
                    extra["side_frames_rewarm_error"] = f"{type(e).__name__}: {e}"[:300]
        if args.graph and hasattr(net, "_autograph") and not args.all_kernel_events:
            # the opt-in a serving loop with a resident batch can make (autograph.py): captured on the caller's own
            # tensors, no input copies, the graph's static outputs handed out.  NOT `value`: run.py hands over other
            # tensors every frame and keeps what it is given.
            rb = _clone(batch, dev)
            net.resident_inputs = net.alias_outputs = True
            try:
                def resident_step():
                    with torch.no_grad():
                        return finish(net(rb))
                for _ in range(4):
                    resident_step()
                torch.cuda.synchronize()
                t_r = bracketed(resident_step, max(n_x, 50))
                extra["resident_batch"] = {"value": N / t_r / 1e6, "ms_per_step": t_r * 1e3,
                                           "what": "same bracket, net.resident_inputs = net.alias_outputs = True: the graph "
                                                   "captured on the caller's resident tensors, no copies in or out (rounds 1-3 "
                                                   "reported this as `value`)"}
            finally:
                net.resident_inputs = net.alias_outputs = False
        torch.cuda.synchronize()

    import gc
    gc.collect()
    gc.disable()                       # a generation-2 collection inside a 1 ms step is a 1-3 ms outlier (timeit does the same)
    ktimer.reset()
    # HIP events around the kernels the roofline objects are built from; --all-kernel-events times every launch
    # (two event records per launch: ~0.3 ms of host time per frame, the frame is then host-bound)
    ktimer.only = None if args.all_kernel_events else ("sweep_variance", "render_rays", "mvs_render", "mvs_sweep", "empty_bracket")
    # The device needs ~20 frames of sustained load to reach its steady state (measured: the first sync-bracketed
    # 512x640 frame after a few tens of ms of idle takes 1.19 ms, the 20th 0.93), and what runs just before this point
    # (the PCIe legs' host copies, captures, the collector above) leaves it idle: an untimed run of the same step brings
    # it to the state of a renderer that has been serving frames, which is what a throughput figure means.  Nothing but
    # the contract's barrier + synchronize sits between it and the timed region.  `--spinup-steps 0` times the cold start;
    # reported as `config.spinup_steps`.
    spinup_steps = args.spinup_steps if (wl["net"] in ("enerf", "boost_enerf") and not wl.get("train")) else 0
    if wl.get("train") and graphed_train is not None:
        # the eager steps in front of the capture and the capturing step itself: untimed, reported like the spin-up
        spinup_steps = max(0, graphed_train.eager_steps + 2 - args.warmup)
    ktimer.enabled = False
    if spinup_steps and not args.pipelined:
        # the same per-step bracket started COLD (the device idle for 50 ms first): what `--spinup-steps 0` would report
        n_c = min(args.steps, 20)
        torch.cuda.synchronize()
        time.sleep(0.05)
        t_c = time.perf_counter()
        for _ in range(n_c):
            step()
            if gather is not None:
                gather.flush()
            torch.cuda.synchronize()
        t_c = (time.perf_counter() - t_c) / n_c
        extra["value_cold"] = {"value": N * (world if args.shard == "views" else 1) / t_c / 1e6, "ms_per_step": t_c * 1e3,
                               "what": f"the timed bracket over {n_c} steps started after 50 ms of idle, no spin-up (rank 0's clock)"}
    for _ in range(spinup_steps):      # a fixed count: every rank issues the same collectives
        step()
        if gather is not None:
            gather.flush()
        torch.cuda.synchronize()
    ktimer.enabled = not args.no_kernel_events
    if step is not eager_step:
        sampled["n"] = 0               # the first timed step carries the brackets
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    step_marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if not wl.get("train") and not args.pipelined:
            if gather is not None:
                gather.flush()
            torch.cuda.synchronize()       # the reference's per-step bracket (run.py:117-123)
            step_marks.append(time.perf_counter())
            if step is eager_step or sampled["evented"]:
                ktimer.collect()           # in-graph event brackets of the replay that just finished
    if gather is not None:
        gather.flush()                 # every exchange issued in the timed region completes inside it
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if args.pipelined:
        ktimer.collect()               # no per-step synchronize: the brackets of the last replay only
    ktimer.enabled = False
    if dev.type == "cuda" and not stub:
        # lost wake-ups of the producer / consumer renderer, sequence faults of the captured frames' feed rings: a timed
        # region that rendered wrong frames must not print a number (the caller is synchronised here anyway)
        from boostmvsnerfs_amd.evaluate import check_device_faults
        check_device_faults(net)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        frames = args.steps * (world if args.shard == "views" else 1)
        value = frames * N / dt / 1e6
        if len(step_marks) > 2:
            # spread of the per-step brackets inside the timed region (`value` is the mean: one host hiccup of a few ms
            # in a 30-step run moves it by 10 %)
            if os.environ.get("BMV_BENCH_DUMP_STEPS"):
                print("[bench] step ms:", " ".join(f"{(b - a) * 1e3:.3f}" for a, b in zip([t0] + step_marks[:-1], step_marks)),
                      file=sys.stderr)
            d = sorted(b - a for a, b in zip([t0] + step_marks[:-1], step_marks))
            extra["step_ms"] = {"min": d[0] * 1e3, "median": d[len(d) // 2] * 1e3, "p90": d[int(len(d) * 0.9)] * 1e3,
                                "max": d[-1] * 1e3, "what": "per-step wall time of the timed region on rank 0"}
        ks = ktimer.summary()
        ks_timed = set(ks)
        for name, v in warm_kernels.items():          # graph mode: the renderer's events come from the eager warm-up
            ks.setdefault(name, v)
        kernels = {name: {"launches": n, "avg_us": mean_ms * 1e3, "min_us": min_ms * 1e3}
                   for name, (n, mean_ms, min_ms) in ks.items()}
        roofline = mfma = None
        if wl["net"] in ("enerf", "boost_enerf"):
            planes = cc.volume_planes
            # ---- roofline of the plane-sweep kernel: both cascade levels, the WORSE one is the headline object
            pmc = {}
            try:    # rocprofv3 --pmc passes on the frame's own sweep inputs (profiles/README.md, scripts/prof_sweep_once.py)
                pmc = json.load(open(os.path.join(REPO, "profiles", "sweep_pmc.json"))).get(args.workload, {})
            except Exception:
                pmc = {}
            levels = {}
            n_src = cfg.enerf.cost_volume_input_views if wl["net"] == "boost_enerf" else wl["views"]
            for lvl, C in ((0, 32), (1, 16)):
                h, w = int(H * cc.volume_scale[lvl]), int(W * cc.volume_scale[lvl])
                name = f"sweep_variance[C={C},D={planes[lvl]},{h}x{w}]"
                if name in kernels:
                    nb = sweep_bytes(n_src, C, int(H * cc.im_feat_scale[lvl]), int(W * cc.im_feat_scale[lvl]), planes[lvl], h, w)
                    k = kernels[name]
                    k.update({"algorithmic_bytes": nb, "GB/s": nb / k["avg_us"] / 1e3})
                    levels[f"level{lvl}"] = {"achieved": k["GB/s"], "frac": k["GB/s"] / HBM_PEAK_GBS, "avg_us": k["avg_us"],
                                             "min_us": k["min_us"], "launches": k["launches"], "algorithmic_bytes": nb,
                                             "traffic": pmc.get(f"sweep_level{lvl}_hbm_bytes"),
                                             "traffic_source": ("profiles/sweep_pmc.json (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE on this "
                                                                "frame's sweep inputs, collected in its own passes, NOT in this run)"
                                                                if pmc.get(f"sweep_level{lvl}_hbm_bytes") else None),
                                             "timed": "HIP events in the timed region" if name in ks_timed else "HIP events in the eager warm-up"}
            if levels:
                worst = min(levels, key=lambda n: levels[n]["frac"])
                L = levels[worst]
                roofline = {"bound": "hbm", "kernel": f"sweep_variance {worst} (a3+a4 fused plane sweep; the worse of the two cascade levels)",
                            "achieved": L["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": L["frac"],
                            "traffic": L["traffic"], "traffic_source": L["traffic_source"],
                            "algorithmic_bytes": L["algorithmic_bytes"], "avg_us": L["avg_us"],
                            "launches": L["launches"], "levels": levels}
                if "empty_bracket" in kernels:
                    # what the same pair of event records reads with NOTHING between them in the same graph: every
                    # bracket includes a share of it (rocprofv3's own begin/end stamps of the sweeps read ~4 us less
                    # than the brackets); reported, not subtracted
                    roofline["event_bracket_empty_us"] = kernels["empty_bracket"]["avg_us"]
            rname = next((n for n in kernels if n.startswith("render_rays[feat=8")), None)
            if rname:
                rays_launch = N // world if (args.shard == "rays" and world > 1) else N
                # the kernel EXECUTES 26.2 kFLOP/sample on the matrix pipe (the view-shared parts of global_fc / color.0
                # are computed once, csrc/mlp.hpp); SURVEY 8(d)'s ALGORITHMIC count of the reference's layers is 50.9.
                # `achieved` / `frac` = executed FLOPs over the kernel's time = the share of the fp32 matrix peak the
                # kernel really uses; the algorithmic figure is a named side field (it can exceed the executed one's
                # roofline share because work was REMOVED, not because the pipe is full)
                ex = 26.2e3 * rays_launch * cc.num_samples[1] / kernels[rname]["avg_us"] / 1e6
                alg = 50.9e3 * rays_launch * cc.num_samples[1] / kernels[rname]["avg_us"] / 1e6
                mfma = {"bound": "mfma", "kernel": "render_rays (a6-a12 fused; MLP: 160 of 206 matrix instructions per tile as "
                                                   "bf16 MFMAs on three-piece fp32 operands, the rest fp32 MFMAs; BMV_RENDER_SPLIT=0: all fp32)",
                        "achieved": ex,
                        "peak": FP32_MFMA_PEAK_TFLOPS, "peak_is": "the fp32 MFMA peak (the rate the same fp32 FLOPs would be bound by on "
                                                                  "fp32 matrix instructions)",
                        "unit": "TFLOP/s", "fp32_equivalent_over_fp32_peak": ex / FP32_MFMA_PEAK_TFLOPS,
                        "flops_counted": "executed (26.2 kFLOP / sample)", "algorithmic_tflops": alg,
                        "algorithmic_over_peak": alg / FP32_MFMA_PEAK_TFLOPS,
                        "avg_us": kernels[rname]["avg_us"], "launches": kernels[rname]["launches"]}
                # `frac` = the share of the kernel's time its matrix pipe is BUSY, each instruction class against its own peak
                # (fp32-equivalent FLOPs over the fp32 peak is not a roofline fraction once 160 of 206 instructions
                # run on the bf16 pipe).  Instruction counts per 32-sample tile from the ISA of render_pc_kernel<2,false,3,true>
                # (S = 3, 8 feature channels): 46 v_mfma_f32_32x32x2_f32 (4 096 FLOP, 64 cycles) + 132 v_mfma_f32_32x32x16_bf16
                # (32 768 FLOP, 32 cycles); all-fp32 form: 206 x 64 cycles.  The rest of the time is vector work, which
                # does not overlap with matrix work on a SIMD (profiles/r5/mfma_valu_coissue.txt).
                if n_src == 3:
                    from boostmvsnerfs_amd import _lib as _bl2
                    split = (_bl2.get_tuning("BMV_RENDER_SPLIT") != 0)
                    tiles = rays_launch * cc.num_samples[1] / 32.0
                    t_s = kernels[rname]["avg_us"] * 1e-6
                    n32, n16 = (46, 132) if split else (206, 0)
                    tf32 = tiles * n32 * 4096 / t_s / 1e12
                    tbf = tiles * n16 * 32768 / t_s / 1e12
                    mfma.update({"frac": tf32 / FP32_MFMA_PEAK_TFLOPS + tbf / BF16_MFMA_PEAK_TFLOPS,
                                 "frac_is": "matrix-pipe busy share of the kernel's time = fp32_pipe.frac + bf16_pipe.frac",
                                 "fp32_pipe": {"achieved": tf32, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": tf32 / FP32_MFMA_PEAK_TFLOPS, "instructions_per_tile": n32},
                                 "bf16_pipe": {"achieved": tbf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": tbf / BF16_MFMA_PEAK_TFLOPS, "instructions_per_tile": n16,
                                               "what": "every one of the six bf16 products of a three-piece fp32 product counted"}})
                else:
                    mfma["frac"] = ex / FP32_MFMA_PEAK_TFLOPS
        else:
            Ns = cc.num_samples[0]
            rname = f"mvs_render[Ns={Ns}]"
            if rname in kernels:
                flops = 251e3 * N * Ns                                 # a25, per launch (one volume)
                tf = flops / kernels[rname]["avg_us"] / 1e6
                mfma = {"bound": "mfma", "kernel": "mvs_render (a21-a25 fused, 6x128 MLP: 15 of its 17 weight chunks as bf16 MFMAs on "
                                                   "three-piece fp32 operands, pts_bias on fp32 MFMAs; BMV_MVS_SPLIT=0: all fp32)",
                        "achieved": tf,
                        "peak": FP32_MFMA_PEAK_TFLOPS, "peak_is": "the fp32 MFMA peak (the rate the same fp32 FLOPs would be bound by on "
                                                                  "fp32 matrix instructions)",
                        "unit": "TFLOP/s", "fp32_equivalent_over_fp32_peak": tf / FP32_MFMA_PEAK_TFLOPS,
                        "flops_counted": "algorithmic (251 kFLOP / sample, SURVEY.md 8(d))",
                        "avg_us": kernels[rname]["avg_us"], "launches": kernels[rname]["launches"]}
                # `frac` = the share of the kernel's time its matrix pipe is BUSY, each instruction class against its own peak
                # (as for the ENeRF renderer above).  Per 32-sample tile (csrc/mvs.hip MvsMlp): split form 40
                # v_mfma_f32_32x32x2_f32 (pts_bias) + 1452 v_mfma_f32_32x32x16_bf16 (6 per bf16 k-step and tile: 121 k-steps
                # x 2 tiles); all-fp32 form 1964 x 64 cycles.
                from boostmvsnerfs_amd import _lib as _bl3
                msplit = (_bl3.get_tuning("BMV_MVS_SPLIT") != 0)
                mtiles = N * Ns / 32.0
                mt_s = kernels[rname]["avg_us"] * 1e-6
                m32, m16 = (40, 1452) if msplit else (1964, 0)
                mtf32 = mtiles * m32 * 4096 / mt_s / 1e12
                mtbf = mtiles * m16 * 32768 / mt_s / 1e12
                mfma.update({"frac": mtf32 / FP32_MFMA_PEAK_TFLOPS + mtbf / BF16_MFMA_PEAK_TFLOPS,
                             "frac_is": "matrix-pipe busy share of the kernel's time = fp32_pipe.frac + bf16_pipe.frac",
                             "fp32_pipe": {"achieved": mtf32, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": mtf32 / FP32_MFMA_PEAK_TFLOPS, "instructions_per_tile": m32},
                             "bf16_pipe": {"achieved": mtbf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": mtbf / BF16_MFMA_PEAK_TFLOPS, "instructions_per_tile": m16,
                                           "what": "every one of the six bf16 products of a three-piece fp32 product counted"}})
            sname = next((n for n in kernels if n.startswith("mvs_sweep[")), None)
            if sname:
                h, w = H // 4, W // 4
                nb = 4 * (3 * 32 * h * w + 2 * 3 * h * w + 41 * Ns * (h + 48) * (w + 48))
                gbs = nb / kernels[sname]["avg_us"] / 1e3
                roofline = {"bound": "hbm", "kernel": "mvs_sweep (a19+a20 padded sweep)", "achieved": gbs,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                            "algorithmic_bytes": nb, "avg_us": kernels[sname]["avg_us"],
                            "launches": kernels[sname]["launches"]}
        headline = args.workload == HEADLINE
        line = {
            "metric": "rendered Mray/s per GPU (512x640, 3 src views, 64 planes)" if headline
            else f"rendered Mray/s per GPU ({args.workload})",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if (args.shard == "views" or world == 1) else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not stub else "synthetic; STUB renderer on CPU + gloo (launcher test, not a measurement)",
            "config": {"workload": args.workload, "network": wl["net"], "H": H, "W": W, "src_views": wl["views"],
                       "volume_planes": list(cc.volume_planes) if "planes" in wl else None,
                       "num_samples": list(cc.num_samples), "render_if": list(cc.render_if),
                       "k_best": wl.get("k_best"), "shard": args.shard if world > 1 else "none",
                       "gather": ("none" if world == 1 or wl.get("train") else
                                  "sync" if args.sync_gather or args.shard == "rays" or not args.pipelined
                                  else "pipelined (1 frame)"),
                       "weights": "random init (seed 0)",
                       "launch": (graph_note if graphed_train is None else
                                  f"forward + loss + backward replayed as one HIP graph, clip + Adam eager ({graphed_train.stats})"),
                       "spinup_steps": spinup_steps,
                       "bracket": ("train step" if wl.get("train") else "pipelined: one synchronize after the K steps" if args.pipelined
                                   else "run.py:117-123: device synchronize after every step")},
            # `value` is the aggregate over all ranks (the metric's "per GPU" names the 1-GPU headline config)
            "value_per_gpu": value / world, "value_extra": extra,
            "roofline": roofline, "roofline_mfma": mfma, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline and (headline or args.cpu_baseline):
            cb = cpu_baseline(args, cfg, wl, sd_cpu, batch_cpu, _SELECTION)
            if cb is not None:
                line["cpu_baseline"] = cb
            line.update(parity_objects(cfg, wl, net, sd_cpu, batch, batch_cpu, split_frame, dev))
            ref_frame = getattr(cpu_baseline, "last_frame", None)
            for name, fr in side_frames.items():
                if ref_frame and name in extra and "error" not in extra[name]:
                    par = {k: _rel_err(fr[k], want)[0] for k, want in ref_frame.items() if k in fr and torch.is_tensor(want)}
                    extra[name]["parity_max_rel"] = max(par.values()) if par else None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
