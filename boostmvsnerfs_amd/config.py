"""`cfg` provider for the network modules.

The reference networks read a module-level yacs singleton built at import time
from sys.argv (lib/config/config.py:8-201).  When this package is dropped into
the reference tree that singleton is used as is:
`get_cfg()` returns `lib.config.cfg` whenever the reference's config module has
been imported.  Stand-alone (bench, tests, GPU box) the same attribute tree is
built here: yaml `parent_cfg` chains (lib/config/config.py:170-188), then the
CLI `opts` list, or one of the built-in presets that restate the values of the
reference yaml files the five BASELINE configs use.

Only the attributes the live lib/networks files read are modelled:
cfg.enerf.{white_bkgd, chunk_size, viewdir_agg, cost_volume_input_views},
cfg.enerf.cas_config.*, cfg.result_dir.
"""
from __future__ import annotations

import ast
import copy
import os
import sys


class CfgNode(dict):
    """Attribute-access dict (the subset of yacs.CfgNode the networks rely on)."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else copy.deepcopy(v)

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("opts must be KEY VALUE pairs")
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = val


# configs/exps/pretrain/enerf/dtu_pretrain.yaml:21-47 (+ defaults of lib/config/config.py)
_ENERF_BASE = {
    "task": "pretrain",
    "exp_name": "enerf",
    "network_module": "lib.networks.enerf.network",
    "result_dir": os.path.join(os.environ.get("workspace", "."), "result"),
    "require_view_selection": False,
    "enerf": {
        "train_input_views": [2, 3, 4],
        "test_input_views": 3,
        "viewdir_agg": True,
        "chunk_size": 1000000,
        "white_bkgd": False,
        "cas_config": {
            "num": 2,
            "depth_inv": [True, False],
            "volume_scale": [0.125, 0.5],
            "volume_planes": [64, 8],
            "im_feat_scale": [0.25, 0.5],
            "im_ibr_scale": [0.25, 1.0],
            "render_scale": [0.25, 1.0],
            "render_im_feat_level": [0, 2],
            "nerf_model_feat_ch": [32, 8],
            "render_if": [True, True],
            "num_samples": [8, 2],
            "num_rays": [4096, 32768],
            "train_img": [True, True],
            "loss_weight": [0.1, 1.0],
        },
    },
}


def _preset(name):
    c = CfgNode(_ENERF_BASE)
    cc = c.enerf.cas_config
    if name == "enerf_pretrain":
        pass
    elif name == "enerf_eval":            # configs/exps/evaluate/enerf/base_eval.yaml:3-5
        cc.render_if = [False, True]
    elif name in ("enerf_ours_eval", "enerf_ours_ft"):
        # configs/exps/pretrain/enerf_ours/dtu_pretrain.yaml:4-15
        c.exp_name = "enerf_ours"
        c.network_module = "lib.networks.boost_enerf.network"
        c.require_view_selection = True
        c.enerf.cost_volume_input_views = 3
        c.enerf.test_input_views = 6
        c.enerf.train_input_views = [5, 6, 7]
        cc.k_best = 4
        # evaluate/enerf_ours/base_eval.yaml renders level 1 only; the fine-tune
        # config (configs/exps/finetune/enerf_ours/free/base.yaml:5-7) renders both.
        cc.render_if = [False, True] if name.endswith("eval") else [True, True]
    elif name in ("mvsnerf_eval", "mvsnerf_ours_eval"):
        # configs/exps/pretrain/mvsnerf/dtu_pretrain.yaml:8-14
        c.exp_name = name.replace("_eval", "")
        c.network_module = "lib.networks.mvsnerf.network"
        cc.num = 1
        cc.depth_inv = [False]
        cc.render_scale = [1.0]
        cc.volume_scale = [0.25]
        cc.num_samples = [32]
        cc.render_if = [True]
        if name == "mvsnerf_ours_eval":
            c.network_module = "lib.networks.boost_mvsnerf.network"
            c.require_view_selection = True
            c.enerf.cost_volume_input_views = 3
            c.enerf.test_input_views = 6
            cc.k_best = 4
    else:
        raise KeyError(f"unknown preset {name!r}")
    return c


PRESETS = ("enerf_pretrain", "enerf_eval", "enerf_ours_eval", "enerf_ours_ft", "mvsnerf_eval", "mvsnerf_ours_eval")


def load_yaml_chain(cfg_file, base=None):
    """yaml with recursive `parent_cfg` inheritance (lib/config/config.py:170-181)."""
    import yaml

    with open(cfg_file, "r") as f:
        cur = yaml.safe_load(f) or {}
    node = base if base is not None else CfgNode(_ENERF_BASE)
    parent = cur.pop("parent_cfg", None)
    if parent:
        node = load_yaml_chain(parent, node)
    node.merge_from_other_cfg(CfgNode(cur))
    return node


def make_cfg(preset="enerf_eval", cfg_file=None, opts=()):
    c = load_yaml_chain(cfg_file) if cfg_file else _preset(preset)
    opts = list(opts)
    if "other_opts" in opts:                       # lib/config/config.py:182-186
        opts = opts[: opts.index("other_opts")]
    c.merge_from_list(opts)
    if "result_dir" in c and "task" in c:          # lib/config/config.py:163
        pass
    return c


_ACTIVE = None


def set_cfg(c):
    """Install the stand-alone cfg the network modules will read."""
    global _ACTIVE
    _ACTIVE = c
    return c


def get_cfg():
    ref = sys.modules.get("lib.config")
    if ref is not None and hasattr(ref, "cfg"):
        return ref.cfg
    global _ACTIVE
    if _ACTIVE is None:
        _ACTIVE = make_cfg("enerf_eval")
    return _ACTIVE


class _CfgProxy:
    """`from boostmvsnerfs_amd.config import cfg` behaves like the reference's
    module-level singleton but always resolves to the active configuration."""

    def __getattr__(self, name):
        return getattr(get_cfg(), name)

    def __getitem__(self, name):
        return get_cfg()[name]

    def __contains__(self, name):
        return name in get_cfg()


cfg = _CfgProxy()
