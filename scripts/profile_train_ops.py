"""Where does a fine-tune step's time go, by torch operator?  Runs bench.py's fine-tune workload under torch.profiler
(one step, shapes recorded) and prints the operators with the most device time plus the shapes behind aten::copy_."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="enerf_ft_512x640_3src")
    ap.add_argument("--stacks", action="store_true")
    ap.add_argument("--rows", type=int, default=40)
    ap.add_argument("--sort", default="self_cuda_time_total")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--workload", a.workload]
    args = bench.parse()
    dev = torch.device("cuda:0")
    cfg, wl, net, sd_cpu, batch_cpu, batch, level = bench.build(args, 0, dev)
    from boostmvsnerfs_amd.train import NetworkWrapper, make_optimizer, train_step
    cc = cfg.enerf.cas_config
    gen = torch.Generator().manual_seed(0)
    for i in range(cc.num):
        batch[f"rgb_{i}"] = torch.rand(1, batch[f"rays_{i}"].shape[1], 3, generator=gen).to(dev)
    net.train()
    wrapper = NetworkWrapper(net)
    opt = make_optimizer(net)
    for _ in range(3):
        train_step(wrapper, opt, batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        train_step(wrapper, opt, batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by=a.sort, row_limit=a.rows, max_name_column_width=60))
    ev = [e for e in prof.key_averages(group_by_input_shape=True)
          if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add_", "aten::add", "aten::fill_", "aten::zero_",
                       "aten::index", "aten::cat", "aten::mul", "aten::sum")]
    ev.sort(key=lambda e: -e.self_device_time_total)
    for e in ev[:45]:
        print(e.key, e.count, e.input_shapes, f"{e.self_device_time_total:.0f}us")


if __name__ == "__main__":
    main()
