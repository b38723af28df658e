"""torch.autograd.Function wrappers: HIP forward + HIP backward for every hot-path op that
carries gradient in fine-tuning (SURVEY.md section 8 "Backward contract", BASELINE config 5).
The training path materialises the per-sample tensors between ops (as the reference does);
inference keeps using the fused kernels.
"""
import torch

from . import ops


class SweepVariance(torch.autograd.Function):
    """a3+a4.  Gradient w.r.t. feats always, w.r.t. depth_values when they require it (level 1)."""

    @staticmethod
    def forward(ctx, feats, proj, depth_values, algo=0):
        ctx.save_for_backward(feats, proj, depth_values)
        return ops.sweep_variance(feats, proj, depth_values, algo=algo)

    @staticmethod
    def backward(ctx, d_var):
        feats, proj, dv = ctx.saved_tensors
        d_feats, d_dv = ops.sweep_variance_bwd(feats, proj, dv, d_var.contiguous(), ctx.needs_input_grad[2])
        return d_feats, None, d_dv, None


class DepthRegress(torch.autograd.Function):
    """a5."""

    @staticmethod
    def forward(ctx, depth_prob, depth_values, depth_inv):
        ctx.save_for_backward(depth_prob, depth_values)
        ctx.inv = bool(depth_inv)
        depth, std = ops.depth_regress(depth_prob, depth_values, depth_inv)
        return depth, std

    @staticmethod
    def backward(ctx, d_depth, d_std):
        prob, vals = ctx.saved_tensors
        d_prob, d_vals = ops.depth_regress_bwd(prob, vals, d_depth.contiguous(), d_std.contiguous(), ctx.inv)
        return d_prob, (d_vals if ctx.needs_input_grad[1] else None), None


class DepthValuesCascade(torch.autograd.Function):
    """a2 (cascade level): depth_values carries gradient to the previous level's depth / std;
    the returned volume bounds are detached (enerf/utils.py:150)."""

    @staticmethod
    def forward(ctx, depth, std, near_far, h, w, D):
        ctx.save_for_backward(depth, std, near_far)
        dv, nf = ops.depth_values_cascade(depth, std, near_far, h, w, D)
        ctx.mark_non_differentiable(nf)
        return dv, nf

    @staticmethod
    def backward(ctx, d_dv, _d_nf):
        depth, std, near_far = ctx.saved_tensors
        d_depth, d_std = ops.depth_values_cascade_bwd(depth, std, near_far, d_dv.contiguous())
        return d_depth, d_std, None, None, None, None


class BuildRays(torch.autograd.Function):
    """a6: gradient flows from the per-ray [near, far] (columns 8:10) to depth and std."""

    @staticmethod
    def forward(ctx, rays, depth, std, near_far, Hr, Wr, depth_inv):
        ctx.save_for_backward(rays, depth, std, near_far)
        ctx.cfg = (int(Hr), int(Wr), bool(depth_inv))
        return ops.build_rays(rays, depth, std, near_far, Hr, Wr, depth_inv)

    @staticmethod
    def backward(ctx, d_rays12):
        rays, depth, std, near_far = ctx.saved_tensors
        Hr, Wr, inv = ctx.cfg
        d_nf = d_rays12[..., 8:10].contiguous()
        d_depth, d_std = ops.build_rays_bwd(rays, depth, std, near_far, d_nf, Hr, Wr, inv)
        return None, d_depth, d_std, None, None, None, None


class SampleAlongDepth(torch.autograd.Function):
    """a7: world_xyz and the normalised depth coordinate carry gradient back to the ray bounds."""

    @staticmethod
    def forward(ctx, rays12, Ns, depth_inv):
        ctx.save_for_backward(rays12)
        ctx.cfg = (int(Ns), bool(depth_inv))
        xyz, uvd, z = ops.sample_along_depth(rays12, Ns, depth_inv)
        ctx.mark_non_differentiable(z)       # z_vals only feed the (detached) depth output
        return xyz, uvd, z

    @staticmethod
    def backward(ctx, d_xyz, d_uvd, _d_z):
        (rays12,) = ctx.saved_tensors
        Ns, inv = ctx.cfg
        d_nf = ops.sample_along_depth_bwd(rays12, d_xyz.contiguous(), d_uvd[..., 2].contiguous(), Ns, inv)
        d_rays = torch.zeros_like(rays12)
        d_rays[..., 8:10] = d_nf
        return d_rays, None, None


class VoxFeat(torch.autograd.Function):
    """a9: gradient to the feature volume (scatter) and to the depth coordinate of uvd."""

    @staticmethod
    def forward(ctx, uvd01, volume, ray_w=0, Ns=0):
        """ray_w / Ns: the samples of uvd01 (B, rays * Ns, 3) are Ns per ray, rays row-major over an image ray_w wide
        (a hint for the backward kernel's tiling, 0 = unknown)."""
        ctx.save_for_backward(uvd01, volume)
        ctx.hints = (int(ray_w or 0), int(Ns or 0))
        return ops.vox_feat(uvd01, volume)

    @staticmethod
    def backward(ctx, d_out):
        uvd01, volume = ctx.saved_tensors
        d_vol, d_d = ops.vox_feat_bwd(uvd01, volume, d_out.contiguous(), *ctx.hints)
        d_uvd = torch.zeros_like(uvd01)
        d_uvd[..., 2] = d_d
        return d_uvd, d_vol, None, None


class ImgFeat(torch.autograd.Function):
    """a10: gradient to the image features (scatter) and to the sample positions."""

    @staticmethod
    def forward(ctx, xyz, img_feat_rgb, src_exts, src_ixts, tar_ext, render_scale, n_grad=None, ray_w=0):
        """n_grad: leading channels of img_feat_rgb that need a gradient (the colour channels appended to the
        features are data); ray_w: the rays of xyz (B, rays, Ns, 3) are row-major over an image this wide (a hint
        for the backward kernel's tiling, 0 = unknown)."""
        ctx.save_for_backward(xyz, img_feat_rgb, src_exts, src_ixts, tar_ext)
        ctx.rs = float(render_scale)
        ctx.hints = (n_grad, int(ray_w or 0))
        return ops.img_feat(xyz, img_feat_rgb, src_exts, src_ixts, tar_ext, render_scale)

    @staticmethod
    def backward(ctx, d_out):
        xyz, img, exts, ixts, tar = ctx.saved_tensors
        n_grad, ray_w = ctx.hints
        d_img, d_xyz = ops.img_feat_bwd(xyz, img, exts, ixts, tar, ctx.rs, d_out.contiguous(), n_grad, ray_w)
        return d_xyz, d_img, None, None, None, None, None, None


class Composite(torch.autograd.Function):
    """a12."""

    @staticmethod
    def forward(ctx, raw, z_vals, white_bkgd=False):
        if white_bkgd:
            raise NotImplementedError("white_bkgd backward")
        ctx.save_for_backward(raw, z_vals)
        rgb, depth, weights = ops.composite(raw, z_vals, False)
        ctx.mark_non_differentiable(weights)
        return rgb, depth, weights

    @staticmethod
    def backward(ctx, d_rgb, d_depth, _d_w):
        raw, z = ctx.saved_tensors
        # no host sync: an all-zero d_depth just costs the kernel its (cheap) softmax pass
        return ops.composite_bwd(raw, z, d_rgb.contiguous(), d_depth.contiguous() if d_depth is not None else None), None, None


class Blend(torch.autograd.Function):
    """a16 with already-normalised masks (built under no_grad, boost_enerf/network.py:142-146)."""

    @staticmethod
    def forward(ctx, raws, masks, z_vals):
        ctx.save_for_backward(raws, masks)
        rgb, depth, weights = ops.blend(raws, masks, z_vals, normalise=False)
        ctx.mark_non_differentiable(depth, weights)
        return rgb, depth, weights

    @staticmethod
    def backward(ctx, d_rgb, _d_depth, _d_w):
        raws, masks = ctx.saved_tensors
        return ops.blend_bwd(raws, masks, d_rgb.contiguous()), None, None


class NerfMLP(torch.autograd.Function):
    """a11.  forward: MFMA kernel.  backward: three launches inside bmv_nerf_mlp_bwd (csrc/mlp_bwd.hip): the data
    path (input gradients; pre-activation gradients and layer inputs parked as per-tile matrices), every weight and
    bias gradient as MFMA products over the sample dimension, and a fixed-order reduction of the per-workgroup partials
    into tensors of the parameters' shapes.  Any number of source views S in {2, 3, 4} (img_feat_rgb_dir (..., S, F + 4)).
    Bit-reproducible (round 5: no atomics left in mlp_bwd.hip; the scatter gradients of VoxFeat / ImgFeat /
    SweepVariance are too under bmv_tuning BMV_DETERMINISTIC, csrc/scatter.hpp)."""

    @staticmethod
    def forward(ctx, vox_feat, img_feat_rgb_dir, feat_ch, *params):
        # params: 16 tensors (weight, bias of the 8 Linear layers in ops.NERF_PARAM_ORDER)
        blob = ops.nerf_pack_weights(list(params), feat_ch)
        ctx.save_for_backward(vox_feat, img_feat_rgb_dir, blob, *params)
        ctx.feat_ch = int(feat_ch)
        return ops.nerf_mlp(vox_feat, img_feat_rgb_dir, blob, feat_ch)

    @staticmethod
    def backward(ctx, d_out):
        vox, img, blob, *params = ctx.saved_tensors
        feat_ch = ctx.feat_ch
        FC = feat_ch + 3
        FCP = 2 * ((FC + 1) // 2)
        blob_bwd = ops.nerf_pack_bwd_weights(list(params), feat_ch)
        lead = vox.shape[:-1]
        S = img.shape[-2]
        voxf, imgf = vox.reshape(-1, 8), img.reshape(-1, S, FC + 4)
        d_vox, d_img, grads = ops.nerf_mlp_bwd(voxf, imgf, d_out.reshape(-1, 4).contiguous(), blob, blob_bwd, feat_ch)
        d_vox_feat = d_vox.t().reshape(*lead, 8)
        d_img_feat = torch.cat([d_img[:, :FC], d_img[:, FCP:FCP + 4]], 1).permute(2, 0, 1).reshape(img.shape)
        return (d_vox_feat, d_img_feat, None) + tuple(grads)


class BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm{2,3}d in training mode, optionally with the ReLU that follows it in ConvBnReLU(3D)
    (lib/networks/enerf/utils.py:10-33), on the HBM-bound kernels of csrc/bn.hip.  Running statistics are updated in
    place exactly as torch does (momentum, unbiased variance)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu):
        y, mean, invstd = ops.bn_train_fwd(x, weight, bias, running_mean, running_var, eps, momentum, relu)
        ctx.save_for_backward(x, y if relu is not False else None, weight, mean, invstd)
        ctx.relu = relu            # False: none, True: ReLU, a float: leaky slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, invstd = ctx.saved_tensors
        dx, dw, db = ops.bn_train_bwd(x, y, dy.contiguous(), weight, mean, invstd, ctx.relu)
        return dx, dw, db, None, None, None, None, None


class MvsSweep(torch.autograd.Function):
    """a19 + a20 (MVSNeRF's padded sweep): gradient to the source features through the masked-variance channels."""

    @staticmethod
    def forward(ctx, imgs_small, feats, proj, depth_values, pad):
        ctx.save_for_backward(feats, proj, depth_values)
        ctx.pad = int(pad)
        return ops.mvs_sweep(imgs_small, feats, proj, depth_values, pad)

    @staticmethod
    def backward(ctx, d_vol):
        feats, proj, dv = ctx.saved_tensors
        return None, ops.mvs_sweep_bwd(feats, proj, dv, d_vol.contiguous(), ctx.pad), None, None, None


class MvsVolFeat(torch.autograd.Function):
    """a22 + a23: the 8 volume channels of the MLP input (values already produced by the render kernel's input
    builder) with their gradient to the regularised volume."""

    @staticmethod
    def forward(ctx, volume, values, rays, src_ext0, src_ixt0, near_far, H, W, pad):
        ctx.save_for_backward(rays, src_ext0, src_ixt0, near_far)
        ctx.cfg = (int(H), int(W), tuple(volume.shape), int(pad))
        return values.clone()

    @staticmethod
    def backward(ctx, d_values):
        rays, e0, k0, nf = ctx.saved_tensors
        H, W, vshape, pad = ctx.cfg
        d_vol = ops.mvs_vol_feat_bwd(rays, e0, k0, nf, d_values.contiguous(), H, W, vshape, pad)
        return d_vol, None, None, None, None, None, None, None, None


class MvsMLP(torch.autograd.Function):
    """a25 Renderer_ours (6 x 128, pts_bias gate, skip concat, alpha / rgb heads) with a HIP backward
    (csrc/mvs_mlp_train.hip): x (npts, 86) + the 22 parameter tensors in bmv_mvs_mlp_params order -> (npts, 4).
    Deterministic.  The forward's activations (1.3 KB x 2508 rows per 32 points) stay on the device until the backward."""

    @staticmethod
    def forward(ctx, x, *params):
        out, act, scratch = ops.mvs_mlp_train_fwd(x, params)
        ctx.save_for_backward(out, *params)
        ctx.act, ctx.scratch = act, scratch
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, *params = ctx.saved_tensors
        if ctx.act is None:
            raise RuntimeError("MvsMLP: the activations of this forward were already consumed by a backward "
                               "(retain_graph / double backward are not supported)")
        dx, grads = ops.mvs_mlp_train_bwd(params, ctx.act, ctx.scratch, out, d_out.contiguous())
        ctx.act = ctx.scratch = None
        return (dx if ctx.needs_input_grad[0] else None,
                *[g if need else None for g, need in zip(grads, ctx.needs_input_grad[1:])])
