"""Where does `value_extra.host_batch_sync` go?  (196.5 -> 85.0 -> 88.7 Mray/s over rounds 3-5, slower than
the eager leg on the same PCIe copies.)  One run.py-style loop (run.py:113-129): every frame the batch is copied from
pinned host memory into NEW device tensors and handed to `net(batch)`; per-iteration wall time, the autograph counters
and the feed ring's fault counter are printed, then the same loop with the capture warmed up first.

    python scripts/probe_host_batch.py [--frames 24]
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from boostmvsnerfs_amd.config import make_cfg, set_cfg  # noqa: E402
from boostmvsnerfs_amd.networks.enerf.network import Network  # noqa: E402
from boostmvsnerfs_amd.synthetic import clone_batch, make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--W", type=int, default=640)
    args = ap.parse_args()
    set_cfg(make_cfg("enerf_eval"))
    torch.manual_seed(0)
    dev = torch.device("cuda")
    net = Network().eval().to(dev)
    batch_cpu = make_batch(args.H, args.W, n_views=3, seed=0)
    batch = clone_batch(batch_cpu, dev)
    N = args.H * args.W
    reads = net._autograph_inputs(batch)
    host = {k: v.cpu().pin_memory() for k, v in batch.items() if torch.is_tensor(v) and (reads is None or k in reads)}
    meta = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
    print("host tensors:", {k: tuple(v.shape) for k, v in host.items()})
    print("keys of the full batch not handed over:", sorted(k for k, v in batch.items() if torch.is_tensor(v) and k not in host))

    def run_py_frame():
        fresh = dict(meta)
        for k, v in host.items():
            fresh[k] = v.to(dev, non_blocking=True)
        with torch.no_grad():
            return net(fresh)

    def loop(tag, frames):
        ag = net._autograph
        for i in range(frames):
            torch.cuda.synchronize()
            before = dict(ag.stats)
            t0 = time.perf_counter()
            t_copy0 = time.perf_counter()
            out = run_py_frame()
            t_host = time.perf_counter() - t_copy0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            delta = {k: ag.stats.get(k, 0) - before.get(k, 0) for k in ag.stats if ag.stats.get(k, 0) != before.get(k, 0)}
            ring = None
            for e in ag.entries.values():
                if e.get("ring") is not None:
                    ring = e["ring"]
            fast = None if ring is None else (ring.fast is not None)
            print(f"[{tag}] frame {i:2d}: {dt * 1e3:8.3f} ms wall ({t_host * 1e3:7.3f} ms until net() returned) = "
                  f"{N / dt / 1e6:7.1f} Mray/s   stats +{delta}   ring.fast={fast}")
            del out
        for e in ag.entries.values():
            if e.get("ring") is not None:
                print(f"[{tag}] feed ring faults: {e['ring'].faults()}")

    loop("cold", args.frames)
    # the headline's resident-ring form on the same network, for scale
    ring3 = [clone_batch(batch_cpu, dev) for _ in range(3)]
    with torch.no_grad():
        for i in range(6):
            net(ring3[i % 3])
    torch.cuda.synchronize()
    ts = []
    for i in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            net(ring3[i % 3])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"[ring of 3 resident device batches] {sum(ts) / len(ts) * 1e3:.3f} ms per frame")
    loop("warm", args.frames)
    # copies alone
    ts = []
    for i in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fresh = {k: v.to(dev, non_blocking=True) for k, v in host.items()}
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del fresh
    print(f"[host->device copies alone] {sum(ts) / len(ts) * 1e3:.3f} ms per frame "
          f"({sum(v.numel() * v.element_size() for v in host.values()) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
