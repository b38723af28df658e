"""Experiment: capture forward + loss + backward of the fine-tune step into one HIP graph (optimizer eager)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    wl_name = sys.argv[1] if len(sys.argv) > 1 else "enerf_ft_512x640_3src"
    sys.argv = [sys.argv[0], "--workload", wl_name]
    args = bench.parse()
    dev = torch.device("cuda:0")
    cfg, wl, net, sd_cpu, batch_cpu, batch, level = bench.build(args, 0, dev)
    from boostmvsnerfs_amd.train import NetworkWrapper, make_optimizer
    cc = cfg.enerf.cas_config
    gen = torch.Generator().manual_seed(0)
    for i in range(cc.num):
        batch[f"rgb_{i}"] = torch.rand(1, batch[f"rays_{i}"].shape[1], 3, generator=gen).to(dev)
    net.train()
    wrapper = NetworkWrapper(net)
    opt = make_optimizer(net)

    def fwd_bwd():
        out, loss, stats, _ = wrapper(batch)
        loss = loss.mean()
        loss.backward()
        return loss

    def finish():
        torch.nn.utils.clip_grad_value_(wrapper.parameters(), 40.0)
        opt.step()

    # eager reference timing
    for _ in range(5):
        opt.zero_grad()
        fwd_bwd()
        finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        opt.zero_grad()
        fwd_bwd()
        finish()
    torch.cuda.synchronize()
    print(f"eager   {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        opt.zero_grad()
        fwd_bwd()
        finish()
        torch.cuda.synchronize()
    print(f"eager, synchronize per step   {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            fwd_bwd()
            finish()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
    torch.cuda.synchronize()
    print("captured", flush=True)
    for _ in range(3):
        g.replay()
        finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
        finish()
    torch.cuda.synchronize()
    print(f"graphed {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step   loss {float(static_loss):.5f}", flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
        finish()
        torch.cuda.synchronize()
    print(f"graphed, synchronize per step {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
