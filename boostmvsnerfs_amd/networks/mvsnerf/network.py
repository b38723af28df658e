"""MVSNeRF on the MI355X hot-path kernels.

Boundary of lib/networks/mvsnerf/network.py:782-1126: `Network()` with sub-modules
`feature`, `cost_reg_2`, `nerf.nerf.*` (identical state-dict keys), `forward(batch)`
-> {rgb,depth,weights}_level0.  The cost volume lives in the frustum of the FIRST of
the three source views (padded by 24 feature pixels), not the target's.

Per cost volume: reference-view projection matrices -> image resize -> fused padded
sweep (reference rgb | warped source rgb | masked variance) -> [3-D regulariser,
csrc/conv.hip engine] -> ONE fused kernel that marches the rays, projects every sample into
the volume and the three images, builds the 86-wide input and runs the 6x128 MLP on
the matrix cores -> compositing kernel.  The reference's 10-chunk Python loop
(network.py:1010-1033) and its per-sample intermediates disappear.  Under autograd (training) the same
kernels build the MLP inputs, the sweep and the volume lookup have HIP backward kernels (csrc/mvs.hip) and the
6 x 128 MLP runs on the layer-wise MFMA kernels with a HIP backward (csrc/mvs_mlp_train.hip, autograd.MvsMLP;
BMV_MVS_MLP_TRAIN=torch keeps nn.Linear + torch autograd).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import autograd as A
from ... import ops
from ... import convnet, switches
from ...config import cfg
from ..enerf.cnn import _Packed, _engine_ok
from ..enerf.conv_train import Conv2d, Conv3d, ConvTranspose3d   # under autograd: engine forward, own weight gradients

PAD = 24   # network.py:1016, 1106


class ABN(nn.Module):
    """Stand-in for inplace_abn.InPlaceABN (batch norm + leaky_relu 0.01) with the same
    state-dict entries (weight, bias, running_mean, running_var); the CUDA extension the
    reference imports does not exist for ROCm."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, slope=0.01):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.eps, self.momentum, self.slope = eps, momentum, slope

    def forward(self, x):
        if self.training and x.is_cuda and x.dtype == torch.float32 and switches.get("BMV_BN") != "torch":
            # batch statistics + leaky ReLU on csrc/bn.hip (as ConvBnReLU's training path, enerf/cnn.py)
            return A.BatchNormTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                          self.momentum, float(self.slope))
        y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training, self.momentum,
                         self.eps)
        return F.leaky_relu(y, self.slope, inplace=True)


class _ConvABN(nn.Module):
    def __init__(self, conv_cls, cin, cout, k, stride, pad):
        super().__init__()
        self.conv = conv_cls(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = ABN(cout)

    def forward(self, x):
        return self.bn(self.conv(x))


def _c2(cin, cout, k=3, stride=1, pad=1):
    return _ConvABN(Conv2d, cin, cout, k, stride, pad)


def _c3(cin, cout, stride=1):
    return _ConvABN(Conv3d, cin, cout, 3, stride, 1)


def _up3(cin, cout):
    return nn.Sequential(ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False), ABN(cout))


class FeatureNet(nn.Module):
    """(B,V,3,H,W) -> (B,V,32,H/4,W/4) (network.py:699-733)."""

    def __init__(self):
        super().__init__()
        self.conv0 = nn.Sequential(_c2(3, 8), _c2(8, 8))
        self.conv1 = nn.Sequential(_c2(8, 16, 5, 2, 2), _c2(16, 16), _c2(16, 16))
        self.conv2 = nn.Sequential(_c2(16, 32, 5, 2, 2), _c2(32, 32), _c2(32, 32))
        self.toplayer = Conv2d(32, 32, 1)
        self._packed = _Packed()

    def _forward_engine(self, x):
        """Inference on the convolution engine (csrc/conv.hip): one launch per conv + ABN block."""
        blocks = [m for seq in (self.conv0, self.conv1, self.conv2) for m in seq]
        P = self._packed.get(self, lambda: [
            *[convnet.pack_conv(*convnet.fold_bn(m.conv.weight, m.bn), stride=m.conv.stride[0]) for m in blocks],
            convnet.pack_conv(self.toplayer.weight, self.toplayer.bias)])
        for m, (wp, bp) in zip(blocks, P):
            x = convnet.conv_fwd(x, wp, bp, m.conv.out_channels, 1, m.conv.kernel_size[0], m.conv.stride[0],
                                 slope=m.bn.slope)
        # the padded sweep reads the map channel-last (a tap = 8 loads of 16 bytes): written so by the 1x1 layer,
        # returned as a (N,32,h,w) VIEW of the (N,h,w,32) buffer (`.contiguous()` gives the reference tensor)
        return convnet.conv_fwd(x, *P[-1], 32, 1, 1, channels_last=True).permute(0, 3, 1, 2)

    def forward(self, x):
        B, V, C, H, W = x.shape
        if _engine_ok(self, x):
            y = self._forward_engine(x.reshape(B * V, C, H, W))            # (B*V, 32, h, w) channel-last strides
            return y.unflatten(0, (B, V))
        y = self.toplayer(self.conv2(self.conv1(self.conv0(x.reshape(B * V, C, H, W)))))
        return y.view(B, V, 32, H // 4, W // 4)


class CostRegNet(nn.Module):
    """3-D U-Net on the 41-channel padded volume -> 8 channels (network.py:735-779)."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv0 = _c3(in_channels, 8)
        self.conv1, self.conv2 = _c3(8, 16, 2), _c3(16, 16)
        self.conv3, self.conv4 = _c3(16, 32, 2), _c3(32, 32)
        self.conv5, self.conv6 = _c3(32, 64, 2), _c3(64, 64)
        self.conv7, self.conv9, self.conv11 = _up3(64, 32), _up3(32, 16), _up3(16, 8)
        self._packed = _Packed()

    def _forward_engine(self, x):
        def build():
            P = {f"conv{i}": convnet.pack_conv(*convnet.fold_bn(getattr(self, f"conv{i}").conv.weight,
                                                                 getattr(self, f"conv{i}").bn),
                                               stride=getattr(self, f"conv{i}").conv.stride[0]) for i in range(7)}
            for name in ("conv7", "conv9", "conv11"):
                up = getattr(self, name)
                P[name] = convnet.pack_convT(*convnet.fold_bn(up[0].weight, up[1], out_dim=1))
            return P
        P = self._packed.get(self, build)
        sl = self.conv0.bn.slope

        def cb(name, t, cout, stride=1):
            return convnet.conv_fwd(t, *P[name], cout, 3, 3, stride, slope=sl)
        s0 = cb("conv0", x, 8)
        s1 = cb("conv2", cb("conv1", s0, 16, 2), 16)
        s2 = cb("conv4", cb("conv3", s1, 32, 2), 32)
        t = cb("conv6", cb("conv5", s2, 64, 2), 64)
        y = convnet.convT3d_fwd(t, *P["conv7"], 32, skip=s2, slope=sl)
        y = convnet.convT3d_fwd(y, *P["conv9"], 16, skip=s1, slope=sl)
        return convnet.convT3d_fwd(y, *P["conv11"], 8, skip=s0, slope=sl)

    def forward(self, x):
        if _engine_ok(self, x):
            return self._forward_engine(x)
        s0 = self.conv0(x)
        s1 = self.conv2(self.conv1(s0))
        s2 = self.conv4(self.conv3(s1))
        y = s2 + self.conv7(self.conv6(self.conv5(s2)))
        y = s1 + self.conv9(y)
        return s0 + self.conv11(y)


def _kaiming(lin):
    nn.init.kaiming_normal_(lin.weight.data)
    nn.init.zeros_(lin.bias.data)
    return lin


class RendererMLP(nn.Module):
    """Parameter holder of Renderer_ours(D=6, W=128, 63/20/3 inputs) (network.py:153-181);
    forward runs the MFMA kernel."""

    def __init__(self, D=6, W=128, input_ch=63, input_ch_views=3, input_ch_feat=20):
        super().__init__()
        if (D, W, input_ch, input_ch_views, input_ch_feat) != (6, 128, 63, 3, 20):
            raise NotImplementedError("the MFMA kernel is laid out for the shipped 6x128 / 63+20+3 renderer")
        self.pts_linears = nn.ModuleList([_kaiming(nn.Linear(input_ch, W))] +
                                         [_kaiming(nn.Linear(W + input_ch if i == 4 else W, W)) for i in range(D - 1)])
        self.pts_bias = nn.Linear(input_ch_feat, W)
        self.views_linears = nn.ModuleList([_kaiming(nn.Linear(input_ch_views + W, W // 2))])
        self.feature_linear = _kaiming(nn.Linear(W, W))
        self.alpha_linear = _kaiming(nn.Linear(W, 1))
        self.rgb_linear = _kaiming(nn.Linear(W // 2, 3))
        self._blob, self._key = None, None

    def _named(self):
        mods = {f"pts_linears.{i}": self.pts_linears[i] for i in range(6)}
        mods.update({"pts_bias": self.pts_bias, "views_linears.0": self.views_linears[0],
                     "feature_linear": self.feature_linear, "alpha_linear": self.alpha_linear,
                     "rgb_linear": self.rgb_linear})
        return mods

    def packed_weights(self):
        mods = self._named()
        key = tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods.values())
        if key != self._key:
            self._blob = ops.mvs_mlp_pack_weights({k: m.weight for k, m in mods.items()},
                                                  {k: m.bias for k, m in mods.items()})
            self._key = key
        return self._blob

    def forward_torch(self, x):
        """Renderer_ours.forward (network.py:201-229) in torch ops (the reference form: tests compare the HIP training
        path with it; BMV_MVS_MLP_TRAIN=torch trains through it); pts_bias multiplies, layer 4's output is concatenated
        behind the embedded point."""
        pts, feat, views = x[..., :63], x[..., 63:83], x[..., 83:86]
        bias = self.pts_bias(feat)
        h = pts
        for i in range(6):
            h = F.relu(self.pts_linears[i](h) * bias)
            if i == 4:
                h = torch.cat([pts, h], -1)
        alpha = F.relu(self.alpha_linear(h))
        h = F.relu(self.views_linears[0](torch.cat([self.feature_linear(h), views], -1)))
        return torch.cat([torch.sigmoid(self.rgb_linear(h)), alpha], -1)

    TRAIN_CHUNK = 1 << 18      # points per autograd.MvsMLP call: 2.6 GB of kept activations

    def _param_list(self):
        """The 22 tensors in bmv_mvs_mlp_params order."""
        m = self._named()
        order = [f"pts_linears.{i}" for i in range(6)]
        rest = ["pts_bias", "views_linears.0", "feature_linear", "alpha_linear", "rgb_linear"]
        return ([m[k].weight for k in order] + [m[k].bias for k in order]
                + [t for k in rest for t in (m[k].weight, m[k].bias)])

    def forward(self, x):
        """x (..., 86) = [embedded ndc 63 | feature 20 | view dir 3] -> (..., 4) = [rgb, alpha].  Under autograd: the
        layer-wise MFMA kernels with a HIP backward (autograd.MvsMLP; BMV_MVS_MLP_TRAIN=torch: nn.Linear + torch
        autograd); otherwise the fused inference kernel."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if not x.is_cuda or switches.get("BMV_MVS_MLP_TRAIN") == "torch":
                return self.forward_torch(x)
            flat = x.reshape(-1, 86)
            params = self._param_list()
            outs = [A.MvsMLP.apply(flat[i:i + self.TRAIN_CHUNK], *params) for i in range(0, flat.shape[0], self.TRAIN_CHUNK)]
            return (outs[0] if len(outs) == 1 else torch.cat(outs)).reshape(*x.shape[:-1], 4)
        return ops.mvs_mlp(x, self.packed_weights())


class MVSNeRF(nn.Module):
    """Wrapper that gives the parameters their `nerf.nerf.*` names (network.py:547-574)."""

    def __init__(self):
        super().__init__()
        self.nerf = RendererMLP()

    def forward(self, x):
        return self.nerf(x)


class VolumeState:
    __slots__ = ("volume", "near_far", "views", "cost_volume")


class Network(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature = FeatureNet()
        self.cost_reg_2 = CostRegNet(32 + 9)
        self.nerf = MVSNeRF()
        self.ray_range = None
        self.capture = None

    # ------------------------------------------------------------------ one cost volume
    def build_volume(self, batch, feats, ids):
        """Sweep + regulariser for the view triplet `ids` (B,3) (network.py:1098-1108,
        boost_mvsnerf/network.py:178-192).  Mutates batch['src_*'] and batch['near_far'] as the reference does."""
        cc = cfg.enerf.cas_config
        D = cc.num_samples[0]
        B = feats.shape[0]
        if B != 1:
            raise NotImplementedError("the MVSNeRF path of the reference is written for batch size 1")
        bi = torch.arange(B, device=ids.device)[:, None]
        rng = batch["depth_ranges"][bi, ids]
        near, far = rng.min() * 0.8, rng.max() * 1.2
        t = torch.linspace(0.0, 1.0, D, device=feats.device, dtype=feats.dtype)
        depth_values = (near * (1.0 - t) + far * t)[None].expand(B, -1).contiguous()
        batch["near_far"] = torch.stack([near, far])
        batch["src_inps"] = batch["all_src_inps"][bi, ids]
        batch["src_exts"] = batch["all_src_exts"][bi, ids]
        batch["src_ixts"] = batch["all_src_ixts"][bi, ids]
        cl = feats.permute(0, 1, 3, 4, 2)
        if cl.is_contiguous():      # engine path: pick the views in the channel-last layout the padded sweep reads
            f = cl[bi, ids].permute(0, 1, 4, 2, 3)
        else:
            f = feats[bi, ids]
        h, w = f.shape[-2:]
        proj = ops.mvs_proj_mats(batch["src_exts"], batch["src_ixts"])
        small = ops.resize_bilinear(batch["src_inps"], h, w)
        vol = (A.MvsSweep.apply(small, f, proj, depth_values, PAD) if torch.is_grad_enabled() and f.requires_grad
               else ops.mvs_sweep(small, f, proj, depth_values, PAD))
        st = VolumeState()
        st.cost_volume = vol if self.capture is not None else None
        st.volume = self.cost_reg_2(vol)[0]
        st.near_far = batch["near_far"]
        st.views = (batch["src_inps"][0], batch["src_exts"][0], batch["src_ixts"][0])
        return st

    def render_volume(self, batch, st, want_mask, outs=None):
        """rays -> raw [rgb, alpha], depths (and visibility masks) for one cost volume (network.py:1003-1042)."""
        cc = cfg.enerf.cas_config
        if cc.render_scale[0] != 1.0:
            raise NotImplementedError("MVSNeRF renders at full resolution in every shipped config")
        rays = batch["rays_0"][0]
        if torch.is_grad_enabled() and (st.volume.requires_grad or any(p.requires_grad for p in self.nerf.parameters())):
            return self._render_volume_train(rays, st, want_mask, outs)
        raw, z, mask, x86 = ops.mvs_render(rays, st.volume, *st.views, st.near_far, self.nerf.nerf.packed_weights(),
                                           Ns=cc.num_samples[0], pad=PAD, want_mask=want_mask,
                                           want_inputs=self.capture is not None, ray_range=self.ray_range, outs=outs)
        if self.capture is not None:
            self.capture.update({"mlp_in": x86, "raw": raw, "cost_volume": st.cost_volume, "volume": st.volume})
        return raw, z, mask

    def _render_volume_train(self, rays, st, want_mask, outs):
        """Differentiable form of render_volume: the render kernel only builds the MLP inputs (marching, NDC point,
        positional encoding, volume / colour lookups, visibility), the 8 volume channels get their gradient to the
        regularised volume (autograd.MvsVolFeat), the MLP runs through RendererMLP.forward (HIP forward + backward under
        autograd: autograd.MvsMLP)."""
        if self.ray_range is not None or outs is not None:
            raise NotImplementedError("training renders every ray of batch['rays_0'] (no ray_range / preallocated outputs)")
        cc = cfg.enerf.cas_config
        src_inps, src_exts, src_ixts = st.views
        H, W = src_inps.shape[-2:]
        _, z, mask, x86 = ops.mvs_render(rays, st.volume.detach(), src_inps, src_exts, src_ixts, st.near_far, None,
                                         Ns=cc.num_samples[0], pad=PAD, want_mask=want_mask, want_inputs=True)
        vf = A.MvsVolFeat.apply(st.volume, x86[..., 63:71], rays, src_exts[0], src_ixts[0], st.near_far, H, W, PAD)
        raw = self.nerf(torch.cat([x86[..., :63], vf, x86[..., 71:]], -1))
        if self.capture is not None:
            self.capture.update({"mlp_in": x86, "raw": raw, "cost_volume": st.cost_volume, "volume": st.volume})
        return raw, z, mask

    @staticmethod
    def ensure_rays(batch):
        """batch['rays_0'] built on the device when the batch does not carry it (ops.make_rays = the full-image branch
        of lib/datasets/enerf_utils.py:25-71; columns 6-7 hold the pixel x, y exactly as the shipped loaders leave
        them -- quirk 9 of SURVEY.md: the reference marches from z = x_pix to z = y_pix)."""
        if "rays_0" not in batch:
            H, W = batch["all_src_inps"].shape[-2:]
            batch["rays_0"] = ops.make_rays(batch["tar_ext"], batch["tar_ixt"], H, W, cfg.enerf.cas_config.render_scale[0])
        return batch

    # reference names kept callable
    def get_proj_mats(self, batch):
        return ops.mvs_proj_mats(batch["src_exts"], batch["src_ixts"])

    def ray_marcher(self, rays, N_sample, **kw):
        raise NotImplementedError("ray marching is fused into the render kernel (ops.mvs_render)")

    # ------------------------------------------------------------------ forward
    def forward(self, batch):
        dev = batch["all_src_inps"].device
        self.ensure_rays(batch)
        feats = self.feature(batch["all_src_inps"])
        B = feats.shape[0]
        ids = torch.tensor([0, 1, 2], device=dev).view(1, 3).expand(B, -1)
        st = self.build_volume(batch, feats, ids)
        raw, z, _ = self.render_volume(batch, st, want_mask=False)
        if self.ray_range is not None:
            raw, z = raw[self.ray_range[0]:self.ray_range[1]], z[self.ray_range[0]:self.ray_range[1]]
        if torch.is_grad_enabled() and raw.requires_grad:
            rgb, depth, weights = A.Composite.apply(raw[None], z[None], cfg.enerf.white_bkgd)
        else:
            rgb, depth, weights = ops.composite(raw[None], z[None], cfg.enerf.white_bkgd)
        return {"rgb_level0": rgb, "depth_level0": depth, "weights_level0": weights}
