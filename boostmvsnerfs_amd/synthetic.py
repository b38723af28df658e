"""Synthetic evaluation batches with the reference's batch layout.

The reference's data loaders (lib/datasets/free/enerf_base.py:73-104,
lib/datasets/free/mvsnerf_base.py:71-99) read images and COLMAP cameras from
disk; neither the datasets nor the checkpoints are available offline, so the
benchmark, the parity tests and the golden-vector generator all use this
recipe instead: a ring of source cameras at z=0 that
look at the point (0, 0, 4), the target camera at the origin looking the same
way, pinhole intrinsics with focal 0.8*W, depth range [2, 8].

Ray layout follows lib/datasets/enerf_utils.py:62-71: rays_i[:, 0:3] is the
camera centre, [:, 3:6] = [x, y, 1] K^-T R_c2w^T (NOT normalised, z-depth
parametrisation), [:, 6:8] the integer pixel (x, y) at render_scale[i].
"""
from __future__ import annotations

import numpy as np
import torch


def look_at_w2c(cam_pos, target, up=(0.0, -1.0, 0.0)):
    """World->camera 4x4 (OpenCV convention: +z forward, +y down)."""
    cam_pos = np.asarray(cam_pos, np.float64)
    fwd = np.asarray(target, np.float64) - cam_pos
    fwd /= np.linalg.norm(fwd)
    up = np.asarray(up, np.float64)
    right = np.cross(fwd, -up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 0)  # rows = camera axes in world coords
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ cam_pos
    return E


def pinhole(H, W):
    return np.array([[0.8 * W, 0, W / 2.0], [0, 0.8 * W, H / 2.0], [0, 0, 1]], np.float64)


def make_rays(tar_ext, tar_ixt, H, W, scale):
    """rays_i of lib/datasets/enerf_utils.py:25-31,62-71 (full-image branch)."""
    K = tar_ixt.copy()
    if scale != 1.0:
        K[:2] *= scale
    h, w = int(H * scale), int(W * scale)
    c2w = np.linalg.inv(tar_ext)
    X, Y = np.meshgrid(np.arange(w), np.arange(h))
    XYZ = np.stack([X, Y, np.ones_like(X)], -1).astype(np.float64)
    XYZ = XYZ @ (np.linalg.inv(K).T @ c2w[:3, :3].T)
    o = np.broadcast_to(c2w[:3, 3], XYZ.shape)
    rays = np.concatenate([o, XYZ, X[..., None], Y[..., None]], -1)
    return rays.astype(np.float32).reshape(-1, 8)


def _image(rng, H, W, phase):
    """Smooth, textured image in [-1, 1]: a few sinusoids plus mild noise."""
    y, x = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    img = np.zeros((3, H, W))
    for c in range(3):
        for k in range(4):
            fx, fy = rng.uniform(1, 9, 2)
            ph = rng.uniform(0, 2 * np.pi) + phase
            img[c] += rng.uniform(0.1, 0.35) * np.sin(2 * np.pi * (fx * x + fy * y) + ph)
    img += 0.05 * rng.standard_normal(img.shape)
    return np.clip(img, -1, 1)


def make_batch(H=512, W=640, n_views=3, render_scales=(0.25, 1.0), seed=0, B=1,
               depth_ranges=False, device="cpu", tar_offset=(0.0, 0.0, 0.0)):
    """Build one evaluation batch.

    n_views == 3 gives the plain ENeRF/MVSNeRF batch; n_views > 3 also fills the
    `all_src_*` keys the boost (multi-cost-volume) networks read.
    """
    rng = np.random.default_rng(seed)
    out = {k: [] for k in ("src_inps", "src_exts", "src_ixts", "tar_ext", "tar_ixt", "near_far")}
    rays = {i: [] for i in range(len(render_scales))}
    for b in range(B):
        K = pinhole(H, W)
        exts, imgs = [], []
        for v in range(n_views):
            ang = 2 * np.pi * (v + 0.25 * b) / n_views + 0.3
            rad = 0.3 + 0.3 * ((v * 7 + 3) % n_views) / max(n_views - 1, 1)
            pos = (rad * np.cos(ang), rad * np.sin(ang), 0.0)
            exts.append(look_at_w2c(pos, (0, 0, 4.0)))
            imgs.append(_image(rng, H, W, 0.2 * v))
        tar_ext = look_at_w2c(tar_offset, (0, 0, 4.0))
        out["src_inps"].append(np.stack(imgs))
        out["src_exts"].append(np.stack(exts))
        out["src_ixts"].append(np.stack([K] * n_views))
        out["tar_ext"].append(tar_ext)
        out["tar_ixt"].append(K)
        out["near_far"].append(np.array([2.0, 8.0]))
        for i, s in enumerate(render_scales):
            rays[i].append(make_rays(tar_ext, K, H, W, s))
    batch = {k: torch.from_numpy(np.stack(v).astype(np.float32)).to(device) for k, v in out.items()}
    for i in rays:
        batch[f"rays_{i}"] = torch.from_numpy(np.stack(rays[i])).to(device)
    for k in ("src_inps", "src_exts", "src_ixts"):
        batch["all_" + k] = batch[k]
    if depth_ranges:
        dr = np.tile(np.array([2.5, 6.5], np.float32), (B, n_views, 1))
        batch["depth_ranges"] = torch.from_numpy(dr).to(device)
    batch["meta"] = {"scene": ["synthetic"] * B, "tar_view": [0] * B, "frame_id": [0] * B}
    return batch


def clone_batch(batch, device=None):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.clone() if device is None else v.to(device).clone()
        elif isinstance(v, dict):
            out[k] = {kk: list(vv) if isinstance(vv, list) else vv for kk, vv in v.items()}
        else:
            out[k] = v
    return out
