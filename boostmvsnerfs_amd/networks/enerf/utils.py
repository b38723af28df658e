"""Functional API of the reference's lib/networks/enerf/utils.py, same names and
argument meaning, each one a thin host wrapper over a HIP kernel of libbmv
(include/bmv.h).  Tensors must live on the GPU; nothing here falls back to
torch ops.  `boost_mvsnerf` star-imports this module in the reference
(lib/networks/boost_mvsnerf/network.py:2), hence `__all__`.
"""
import torch

from ... import ops
from ...config import cfg

__all__ = ["get_proj_mats", "homo_warp", "get_depth_values", "build_feature_volume", "depth_regression",
           "build_rays", "sample_along_depth", "get_vox_feat", "get_img_feat", "get_ndc_coords", "mask_viewport", "raw2outputs",
           "raw2outputs_blend", "unpreprocess"]


def _cas():
    return cfg.enerf.cas_config


def get_proj_mats(batch, src_scale, tar_scale):
    """utils.py:35-55 -> (B,S,3,4)."""
    return ops.proj_mats(batch["src_exts"], batch["src_ixts"], batch["tar_ext"], batch["tar_ixt"], src_scale,
                         tar_scale)


def homo_warp(src_feat, proj_mat, depth_values):
    """utils.py:57-95 -> (warped (B,C,D,h,w), grid (B,D,h,w,2))."""
    return ops.homo_warp(src_feat, proj_mat, depth_values, want_grid=True)


def get_depth_values(batch, D, level, device, depth, std, near_far):
    """utils.py:98-153 -> (depth_values (B,D,h,w), near_far (B,2,h,w))."""
    cc = _cas()
    H, W = batch["src_inps"].shape[-2:]
    h, w = int(H * cc.volume_scale[level]), int(W * cc.volume_scale[level])
    if depth is None:
        return ops.depth_values_uniform(batch["near_far"], D, h, w, cc.depth_inv[level])
    if not cc.depth_inv[level - 1] or cc.depth_inv[level]:
        # utils.py:129-144 holds an ipdb trap and a shape bug: unreachable with shipped configs
        raise NotImplementedError("cascade levels must go disparity -> depth (depth_inv=[True, False, ...])")
    return ops.depth_values_cascade(depth, std, near_far, h, w, D)


def build_feature_volume(feature, batch, D, depth, std, near_far, level, algo=0):
    """utils.py:324-351 -> (variance volume (B,C,D,h,w), depth_values, near_far)."""
    cc = _cas()
    depth_values, near_far = get_depth_values(batch, D, level, feature.device, depth, std, near_far)
    proj = get_proj_mats(batch, cc.im_feat_scale[level], cc.volume_scale[level])
    return ops.sweep_variance(feature, proj, depth_values, algo=algo), depth_values, near_far


def depth_regression(depth_prob, depth_values, level, batch=None):
    """utils.py:722-731 (the level == -1 branch :681-720 is dead code)."""
    if level < 0:
        raise NotImplementedError("depth_regression(level=-1) is dead code in the reference")
    return ops.depth_regress(depth_prob, depth_values, _cas().depth_inv[level])


def build_rays(depth, std, batch, training, near_far, level, up_scale=2.0):
    """utils.py:392-422 -> rays (B,N,12)."""
    cc = _cas()
    H, W = batch["src_inps"].shape[-2:]
    rs = cc.render_scale[level]
    return ops.build_rays(batch[f"rays_{level}"], depth, std, near_far, int(H * rs), int(W * rs),
                          cc.depth_inv[level])


def sample_along_depth(rays, N_samples, level):
    """utils.py:424-443 -> world_xyz (B,N,Ns,3), uvd (B,N,Ns,3), z_vals (B,N,Ns)."""
    return ops.sample_along_depth(rays, N_samples, _cas().depth_inv[level])


def get_vox_feat(ndc_xyz, feature_volume):
    """utils.py:458-460; ndc_xyz (B,P,3) in [0,1]^3 -> (B,P,C)."""
    return ops.vox_feat(ndc_xyz, feature_volume)


def get_img_feat(xyz, img_feat_rgb, batch, training, level):
    """utils.py:753-786 -> (B, N*Ns, S, C+4)."""
    return ops.img_feat(xyz, img_feat_rgb, batch["src_exts"], batch["src_ixts"], batch["tar_ext"],
                        _cas().render_scale[level])


def _inv_scale(inv_scale):
    if torch.is_tensor(inv_scale):
        return tuple(float(v) for v in inv_scale.reshape(-1, 2)[0].tolist())
    return inv_scale


def get_ndc_coords(world_xyz, src_ext, src_ixt, inv_scale):
    """utils.py:490-508: world_xyz (B,N,Ns,3), ONE source view per item -> (B,N,Ns,3) = (u, v, z)."""
    inv_w, inv_h = _inv_scale(inv_scale)
    return ops.ndc_coords(world_xyz, src_ext, src_ixt, inv_w, inv_h)


def mask_viewport(world_xyz, src_exts, src_ixts, inv_scale):
    """utils.py:510-520 -> (B, N*Ns, 1).  inv_scale: (W-1, H-1) as a pair of floats or a (B,2) tensor."""
    if torch.is_tensor(inv_scale):
        inv_w, inv_h = (float(v) for v in inv_scale.reshape(-1, 2)[0].tolist())
    else:
        inv_w, inv_h = inv_scale
    return ops.mask_viewport(world_xyz, src_exts, src_ixts, inv_w, inv_h)[..., None]


def raw2outputs(raw, z_vals, white_bkgd=False):
    """utils.py:605-637; `weights` are the softmaxed ones, as in the reference."""
    rgb, depth, weights = ops.composite(raw, z_vals, white_bkgd)
    return {"rgb": rgb, "depth": depth, "weights": weights}


def raw2outputs_blend(raws, masks, z_vals, white_bkgd=False):
    """utils.py:639-667; masks already normalised over K."""
    if white_bkgd:
        raise NotImplementedError           # as the reference, utils.py:660-661
    B, K, N, Ns = raws.shape[:4]
    rgb, depth, weights = ops.blend(raws, masks.reshape(B, K, N, Ns), z_vals, normalise=False)
    return {"rgb": rgb, "depth": depth, "weights": weights}


def unpreprocess(data, shape=(1, 1, 3, 1, 1), render_scale=1.0):
    """utils.py:669-676: [-1,1] -> [0,1] and bilinear (align_corners) resize by render_scale."""
    H, W = data.shape[-2:]
    return ops.unpreprocess(data, int(H * render_scale), int(W * render_scale))
