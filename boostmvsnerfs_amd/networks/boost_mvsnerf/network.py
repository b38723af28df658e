"""BoostMVSNeRFs on the MVSNeRF backbone (lib/networks/boost_mvsnerf/network.py:11-211):
K padded cost volumes, each in the frustum of its own first source view, rendered with
the fused sample+MLP kernel and fused by the visibility-weighted blend kernel.  View
selection needs no network at all (128 marched samples + viewport test per triplet)."""
import json
import os

import torch

from ... import ops
from ...config import cfg
from ..boost_enerf.network import greedy_cover, view_triplets
from ..mvsnerf import network as mvsnerf_network

N_MARCH = 128   # boost_mvsnerf/network.py:29-30


class Network(mvsnerf_network.Network):
    def __init__(self, preprocess=False):
        super().__init__()
        self.view_selection_outputs = None
        if not preprocess:
            path = os.path.join(cfg.result_dir, "view_selection.json")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: run the view-selection preprocess first")
            with open(path, "r") as f:
                self.view_selection_outputs = json.load(f)

    # ------------------------------------------------------------------ view selection
    def calc_mask(self, src_views_id, batch):
        """(B, N) visibility of one triplet along every target ray (network.py:23-45)."""
        ids = torch.as_tensor(src_views_id, device=batch["all_src_inps"].device)
        H, W = batch["all_src_inps"].shape[-2:]
        rs = cfg.enerf.cas_config.render_scale[0]
        z, vis = ops.mvs_march_mask(batch["rays_0"][0], batch["all_src_exts"][0, ids], batch["all_src_ixts"][0, ids],
                                    N_MARCH, int(W * rs) - 1, int(H * rs) - 1)
        m = (vis / N_MARCH)[None, ..., None].expand(-1, -1, -1, 4).contiguous()
        rgb, _, _ = ops.composite(m, z[None], cfg.enerf.white_bkgd)
        return {"mask_level0": rgb.mean(-1)}

    def forward_view_selection(self, batch):
        N = batch["all_src_inps"].shape[1]
        with torch.no_grad():
            masks = torch.stack([self.calc_mask(ids, batch)["mask_level0"] for ids in view_triplets(N, 3)])
        sel = greedy_cover(masks.reshape(masks.shape[0], 1, -1), cfg.enerf.cas_config.k_best)
        key = f"{batch['meta']['scene'][0]}_{batch['meta']['tar_view'][0]}"
        return {key: [int(s) for s in sel]}

    # ------------------------------------------------------------------ fused forward
    def forward(self, batch):
        if self.view_selection_outputs is None:
            raise RuntimeError("Network(preprocess=True) only supports forward_view_selection()")
        cc = cfg.enerf.cas_config
        self.ensure_rays(batch)
        dev = batch["all_src_inps"].device
        B, N = batch["all_src_inps"].shape[:2]
        trip = torch.tensor(view_triplets(N, cfg.enerf.cost_volume_input_views), device=dev)
        picks = [self.view_selection_outputs[f"{s}_{v}"] for s, v in zip(batch["meta"]["scene"], batch["meta"]["tar_view"])]
        if any(not (0 <= int(v) < trip.shape[0]) for row in picks for v in row):
            raise ValueError(f"view_selection.json holds a triplet index outside [0, {trip.shape[0]}) for {N} source views")
        k_best = torch.tensor(picks, device=dev)
        K = int(cc.k_best)
        if k_best.shape[1] < K:
            raise ValueError(f"view_selection.json holds {k_best.shape[1]} volumes per target, cfg k_best={K}")
        sel = trip[k_best[:, :K]]                                   # (B,K,3)
        feats = self.feature(batch["all_src_inps"])
        n_rays, Ns = batch["rays_0"].shape[1], cc.num_samples[0]
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if train:      # differentiable path: per-volume tensors are stacked, nothing is written in place
            parts = []
            for k in range(K):
                st = self.build_volume(batch, feats, sel[:, k])
                parts.append(self.render_volume(batch, st, want_mask=True))
            raws = torch.stack([p[0] for p in parts])[None]
            zs = torch.stack([p[1] for p in parts])[None]
            ms = torch.stack([p[2] for p in parts])[None]
        else:
            raws = torch.empty(1, K, n_rays, Ns, 4, device=dev)
            zs = torch.empty(1, K, n_rays, Ns, device=dev)
            ms = torch.empty(1, K, n_rays, Ns, device=dev)
            for k in range(K):
                st = self.build_volume(batch, feats, sel[:, k])
                self.render_volume(batch, st, want_mask=True, outs=(raws[0, k], zs[0, k], ms[0, k]))
        if self.ray_range is not None:
            b, e = self.ray_range
            raws, zs, ms = raws[:, :, b:e].contiguous(), zs[:, :, b:e].contiguous(), ms[:, :, b:e].contiguous()
        if self.capture is not None:
            self.capture.update({"raws": raws, "zs": zs, "masks": ms})
        if cfg.enerf.white_bkgd:
            raise NotImplementedError
        if train:      # masks are constants of the geometry (boost_mvsnerf/network.py:97-135): normalise, then blend
            from ...autograd import Blend
            with torch.no_grad():
                tot = ms.sum(1, keepdim=True)
                ms = torch.where(tot > 0, ms / tot, torch.full_like(ms, 1.0 / ms.shape[1]))
            rgb, depth, weights = Blend.apply(raws, ms, zs)
        else:
            rgb, depth, weights = ops.blend(raws, ms, zs, normalise=True)
        return {"rgb_level0": rgb, "depth_level0": depth, "weights_level0": weights}
