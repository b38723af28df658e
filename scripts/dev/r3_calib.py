"""Calibration prints for two test tolerances (boost visibility flips; embedding spread)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from conftest import load_fixture, tiny_cfg
DEV = "cuda"

def flips():
    import json, tempfile
    from boostmvsnerfs_amd.config import set_cfg
    fx = load_fixture("boost_enerf_tiny")
    c = tiny_cfg(fx, "enerf_ours_eval")
    d = tempfile.mkdtemp()
    c.result_dir = d
    with open(os.path.join(d, "view_selection.json"), "w") as f:
        json.dump(json.loads(str(fx.raw["extra/view_selection_json"])) if "extra/view_selection_json" in fx.raw else {}, f)
    print("keys", [k for k in fx.raw if k.startswith("extra/")])

def embedding():
    from oracle import mvsnerf as M
    fx = load_fixture("mvsnerf_tiny")
    caps = sorted(k for k in fx.raw if "run_network_mvs" in k)
    x = torch.cat([torch.from_numpy(fx.raw[k]).reshape(-1, 86) for k in caps if fx.raw[k].shape[-1] == 86])
    ndc = x[:, :3]
    e32 = M.embed(ndc.float())
    e64 = M.embed(ndc.double())
    print("embed fp32 vs fp64 spread: max", float((e32.double() - e64).abs().max()), "rms(e64)", float(e64.pow(2).mean().sqrt()))
    print("reference capture vs fp64 of its own ndc: max", float((x[:, :63].double() - e64).abs().max()))

embedding()
