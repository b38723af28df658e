"""The reference's evaluate driver around `Network.forward` (run.py:71-129), for batches the caller supplies.

Scope (SURVEY.md section 8b, item ii): the timing bracket that DEFINES the headline metric and the
view-selection bootstrap of the K-volume networks -- not the data loaders or the evaluators of the reference
(out of scope; `evaluate` takes any iterable of batch dicts and an optional `on_output` callback in the evaluator's
place).

    run.py:88-92    view_selection.json missing -> run the preprocess network over the batches, dump the json
    run.py:71-85    get_view_selection: batch to the GPU, synchronize, forward_view_selection, synchronize
    run.py:113-123  per batch: tensors to the GPU, synchronize, t0, network(batch), synchronize, t1
    run.py:126-129  FPS = 1 / mean(net_time[1:])  (the first iteration is dropped when there is more than one)
"""
from __future__ import annotations

import json
import os
import time

import torch


def to_cuda(batch, device="cuda"):
    """run.py:114-116: every entry but 'meta' goes to the GPU."""
    return {k: (v.to(device) if (k != "meta" and torch.is_tensor(v)) else v) for k, v in batch.items()}


def get_view_selection(batches, network):
    """run.py:71-85."""
    outputs = {}
    for batch in batches:
        batch = to_cuda(batch)
        with torch.no_grad():
            torch.cuda.synchronize()
            output = network.forward_view_selection(batch)
            torch.cuda.synchronize()
        outputs.update(output)
    return outputs


def ensure_view_selection(cfg, make_preprocess_network, batches):
    """run.py:88-92 + 39-69: if `<cfg.result_dir>/view_selection.json` does not exist, build it with the preprocess
    network (`make_preprocess_network()` -> Network(preprocess=True) on the GPU, eval mode).  Returns the path."""
    path = os.path.join(cfg.result_dir, "view_selection.json")
    if cfg.get("require_view_selection") and not os.path.exists(path):
        outputs = get_view_selection(batches, make_preprocess_network())
        os.makedirs(cfg.result_dir, exist_ok=True)
        with open(path, "w") as f:
            json.dump(outputs, f)
    return path


def evaluate(network, batches, on_output=None, graph=None):
    """run.py:113-129.  Returns {'net_time': [...], 'FPS': ..., 'Mray/s': ...} with the reference's definition of FPS
    (mean over iterations 2..n when n > 1) and the rays of the last rendered level per second next to it.

    The bracket is around `network(batch)` and nothing else.  The networks of this package replay a HIP graph of the
    frame from the second call with the same shapes on (autograph.AutoGraph, inside `forward`): the iteration that
    triggers a capture pays for it inside its own bracket, the batch's tensors are copied into the graph's private
    input buffers inside the bracket too (one multi-tensor launch), and the outputs handed to `on_output` are fresh
    tensors like the reference's (`network.alias_outputs = True` hands out the graph's static tensors instead: consume
    them before the next iteration).  graph=False times the eager launches instead
    (`network._forward_checked`); `stats` reports how many iterations were eager / captures / replays."""
    network.eval()
    net_time, rays = [], 0
    call = network
    if graph is False and hasattr(network, "_forward_checked"):
        call = network._forward_checked
    for it, batch in enumerate(batches):
        batch = to_cuda(batch)
        with torch.no_grad():
            torch.cuda.synchronize()
            start = time.time()
            output = call(batch)
            torch.cuda.synchronize()
            end = time.time()
        net_time.append(end - start)
        rgb = [v for k, v in sorted(output.items()) if k.startswith("rgb_level")][-1]
        rays = rgb.shape[0] * rgb.shape[1]
        if on_output is not None:
            on_output(output, batch)
    mean = sum(net_time[1:]) / (len(net_time) - 1) if len(net_time) > 1 else net_time[0]
    res = {"net_time": net_time, "FPS": 1.0 / mean, "Mray/s": rays / mean / 1e6}
    ag = getattr(network, "_autograph", None)
    if ag is not None:
        res["stats"] = dict(ag.stats)
    check_device_faults(network)
    return res


def check_device_faults(network=None):
    """The two device-side fault counters of the inference path, read where the caller is synchronised anyway: lost
    wake-ups of the producer / consumer renderer (bmv_render_pc_check: frames with unwritten pixels) and, for a network
    that replays captured frames, sequence mismatches of their feed rings (a frame that ran on another call's
    pointers).  Raises instead of handing back silently wrong frames."""
    from . import _lib
    _lib.check(_lib.load().bmv_render_pc_check(1), "render_pc_check")
    ag = getattr(network, "_autograph", None) if network is not None else None
    if ag is not None:
        ag.check_faults()
