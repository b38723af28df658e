#!/usr/bin/env python3
"""Benchmark of the BoostMVSNeRFs rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): rendered Mray/s (512x640 target, 3 source views, 64 depth
planes) = rays rendered by the whole job per second of `Network.forward(batch)`,
the sync-bracketed call the reference times in run.py:117-123.  A step is one
forward over one synthetic batch already resident in HBM.

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
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (preset, H, W, volume_planes, n_views)
    "enerf_512x640_3src_64planes": ("enerf_eval", 512, 640, [64, 8], 3),      # BASELINE configs[1] (metric config)
    "enerf_256x320_3src_32planes": ("enerf_eval", 256, 320, [32, 8], 3),      # BASELINE configs[0]
}


def sweep_bytes(S, C, Hs, Ws, D, h, w):
    """read every source feature map once + write the variance volume once."""
    return 4 * (S * C * Hs * Ws + C * D * h * w)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="enerf_512x640_3src_64planes", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="views", choices=["views", "rays"])
    ap.add_argument("--sweep-algo", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    return ap.parse_args()


def cpu_baseline(cfg, state_dict, batch_cpu, H, W):
    """The oracle (CPU port of the reference path) on this host's cores: one full frame."""
    from oracle import enerf as O   # checker / baseline only
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    with torch.no_grad():
        O.enerf_forward(state_dict, make_batch(64, 96), cfg)              # page-in / thread-pool warm-up
        t0 = time.perf_counter()
        O.enerf_forward(state_dict, clone_batch(batch_cpu), cfg)
        dt = time.perf_counter() - t0
    return {"value": H * W / dt / 1e6, "unit": "Mray/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 frame {H}x{W} (whole workload), oracle/enerf.py torch-CPU fp32, {dt:.2f} s",
            "host_cpus": os.cpu_count()}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", init_method="env://", rank=rank, world_size=world, device_id=dev)

    from boostmvsnerfs_amd import ktimer, sharding
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch

    preset, H, W, planes, n_views = WORKLOADS[args.workload]
    cfg = make_cfg(preset)
    cfg.enerf.cas_config.volume_planes = list(planes)
    set_cfg(cfg)
    cc = cfg.enerf.cas_config

    torch.manual_seed(0)
    net = Network().eval()
    sd_cpu = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    net.sweep_algo = args.sweep_algo
    # views sharding: rank r renders its own target (camera shifted along x), same sources
    tar_offset = (0.05 * rank, 0.0, 0.0) if args.shard == "views" else (0.0, 0.0, 0.0)
    batch_cpu = make_batch(H, W, n_views=n_views, seed=0, tar_offset=tar_offset)
    batch = clone_batch(batch_cpu, dev)
    N = H * W
    if args.shard == "rays" and world > 1:
        net.ray_range = sharding.ray_slice(N, world, rank)
    gather = sharding.TileGather(world, N if args.shard == "views" else None, dev) if world > 1 else None

    def step():
        with torch.no_grad():
            out = net(batch)
        if gather is not None:
            if args.shard == "views":
                return gather.all_gather_frames(out["rgb_level1"], out["depth_level1"])
            return gather.all_gather_ray_tiles(out["rgb_level1"], out["depth_level1"], N)
        return out

    for _ in range(args.warmup):
        step()
    ktimer.reset()
    ktimer.enabled = not args.no_kernel_events
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ktimer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        frames = args.steps * (world if args.shard == "views" else 1)
        value = frames * N / dt / 1e6
        # ---- roofline of the plane-sweep kernel (level 1 launch: the larger one)
        ks = ktimer.summary()
        h1, w1 = int(H * cc.volume_scale[1]), int(W * cc.volume_scale[1])
        h0, w0 = int(H * cc.volume_scale[0]), int(W * cc.volume_scale[0])
        lv = {
            0: (f"sweep_variance[C=32,D={planes[0]},{h0}x{w0}]",
                sweep_bytes(3, 32, int(H * cc.im_feat_scale[0]), int(W * cc.im_feat_scale[0]), planes[0], h0, w0)),
            1: (f"sweep_variance[C=16,D={planes[1]},{h1}x{w1}]",
                sweep_bytes(3, 16, int(H * cc.im_feat_scale[1]), int(W * cc.im_feat_scale[1]), planes[1], h1, w1)),
        }
        kernels = {}
        for lvl, (name, nbytes) in lv.items():
            if name in ks:
                n, mean_ms, min_ms = ks[name]
                kernels[f"sweep_level{lvl}"] = {"launches": n, "avg_us": mean_ms * 1e3, "min_us": min_ms * 1e3,
                                                "algorithmic_bytes": nbytes, "GB/s": nbytes / (mean_ms * 1e-3) / 1e9}
        for name, (n, mean_ms, min_ms) in ks.items():
            if not name.startswith("sweep_variance"):
                kernels[name] = {"launches": n, "avg_us": mean_ms * 1e3, "min_us": min_ms * 1e3}
        rname = f"render_rays[feat=8,Ns={cc.num_samples[1]},mode=0]"
        mfma = None
        if rname in ks:
            n, mean_ms, min_ms = ks[rname]
            rays_launch = N // world if (args.shard == "rays" and world > 1) else N
            flops = 50.9e3 * rays_launch * cc.num_samples[1]          # SURVEY.md 8(d) algorithmic FLOPs of a11
            mfma = {"bound": "mfma", "kernel": "render_rays (a6-a12 fused, fp32 MFMA MLP)", "launches": n,
                    "avg_us": mean_ms * 1e3, "achieved": flops / (mean_ms * 1e-3) / 1e12,
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / (mean_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
        roofline = None
        if "sweep_level1" in kernels:
            k = kernels["sweep_level1"]
            traffic = None
            pmc = os.path.join(REPO, "profiles", "sweep_pmc.json")   # rocprofv3 --pmc pass, see profiles/README.md
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(args.workload, {}).get("sweep_level1_hbm_bytes")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": "sweep_variance level 1 (a3+a4 fused plane sweep)",
                        "achieved": k["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["GB/s"] / HBM_PEAK_GBS,
                        "traffic": traffic, "algorithmic_bytes": k["algorithmic_bytes"], "avg_us": k["avg_us"],
                        "launches": k["launches"]}
        line = {
            "metric": "rendered Mray/s per GPU (512x640, 3 src views, 64 planes)" if "512x640" in args.workload
            else "rendered Mray/s per GPU", "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.shard == "views" else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": args.workload, "network": "enerf", "H": H, "W": W,
                                            "src_views": n_views, "volume_planes": planes,
                                            "num_samples": list(cc.num_samples), "render_if": list(cc.render_if),
                                            "shard": args.shard if world > 1 else "none",
                                            "weights": "random init (seed 0)"},
            "roofline": roofline, "roofline_mfma": mfma, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd_cpu, batch_cpu, H, W)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
