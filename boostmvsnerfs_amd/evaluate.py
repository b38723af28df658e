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


def _graph_key(batch, network):
    """What a captured frame is specialised to beyond tensor VALUES: shapes, and -- for the K-volume networks, whose
    forward looks its cost volumes up in view_selection.json by (scene, target view) -- the selected triplets."""
    shapes = tuple((k, tuple(v.shape)) for k, v in sorted(batch.items()) if torch.is_tensor(v))
    sel = None
    table = getattr(network, "view_selection_outputs", None)
    if table is not None:
        meta = batch["meta"]
        sel = tuple(tuple(table[f"{s}_{v}"]) for s, v in zip(meta["scene"], meta["tar_view"]))
    return shapes, sel


def evaluate(network, batches, on_output=None, graph=False):
    """run.py:113-129.  Returns {'net_time': [...], 'FPS': ..., 'Mray/s': ...} with the reference's definition of FPS
    (mean over iterations 2..n when n > 1) and the rays of the last rendered level per second next to it.

    graph=True: after the first (eager) iteration the forward is captured into a HIP graph (framegraph.FrameGraph) on
    static copies of the batch; later batches of the same shapes (and, for the K-volume networks, the same selected
    triplets) are copied into those buffers INSIDE the timed bracket and replayed -- the host cost of a frame drops from
    ~45 launches to one; anything else falls back to the eager call.  The outputs handed to `on_output` are then the
    graph's static tensors: consume them before the next iteration."""
    network.eval()
    net_time, rays = [], 0
    fg, fg_key, static = None, None, None
    for it, batch in enumerate(batches):
        batch = to_cuda(batch)
        key = _graph_key(batch, network) if graph else None
        if graph and it > 0 and (fg is None or key != fg_key):
            from .framegraph import FrameGraph
            static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            # forward may ADD keys to the batch (rays built on the device from the target camera): give every call a
            # fresh shallow copy, so that such tensors are rebuilt inside the captured frame instead of being baked in
            fg, fg_key = FrameGraph(lambda b: network(dict(b)), static, cut=None), key   # untimed, like a warm-up iteration
        with torch.no_grad():
            torch.cuda.synchronize()
            start = time.time()
            if fg is not None and key == fg_key:
                for k, v in batch.items():
                    if torch.is_tensor(v):
                        static[k].copy_(v)
                output = fg.replay()
            else:
                output = network(batch)
            torch.cuda.synchronize()
            end = time.time()
        net_time.append(end - start)
        rgb = [v for k, v in sorted(output.items()) if k.startswith("rgb_level")][-1]
        rays = rgb.shape[0] * rgb.shape[1]
        if on_output is not None:
            on_output(output, batch)
    mean = sum(net_time[1:]) / (len(net_time) - 1) if len(net_time) > 1 else net_time[0]
    return {"net_time": net_time, "FPS": 1.0 / mean, "Mray/s": rays / mean / 1e6}
