"""2-D feature pyramid and 3-D cost regularisers of ENeRF.

These convolution stacks sit between the hot-path kernels.  The modules keep torch
parameters (so checkpoints load and training works through autograd), but inference runs them on the package's own
implicit-GEMM MFMA convolution engine (csrc/conv.hip via convnet.py) with eval-mode batch norm folded into the
weights.  Under autograd the same engine does the convolution forward and the 3-D data gradients (weights repacked on
the device every step), csrc/conv_wgrad.hip the weight gradients and csrc/bn.hip the training-mode batch norm;
MIOpen is left with the data gradients of the two stride-2 5x5 layers (conv_train.py).
Module/parameter names reproduce the reference's state-dict keys exactly
(lib/networks/enerf/feature_net.py:4-36, cost_reg_net.py:4-86, utils.py:10-33)
so `load_state_dict(ckpt['net'], strict=True)` accepts reference checkpoints.
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import autograd as A
from ... import convnet, ops, switches
from .conv_train import Conv2d, Conv3d, ConvTranspose3d, _Conv3dFn   # engine forward / data gradients, MFMA weight gradients


def engine_ok(module, x):
    return _engine_ok(module, x)


def _engine_ok(module, x):
    """Inference (eval-mode batch norm, no autograd) on the GPU runs on the convolution engine
    (csrc/conv.hip); training keeps the torch modules (MIOpen forward / data gradients).
    CPU tensors are refused like everywhere else on the path: there is no CPU fallback."""
    if not x.is_cuda:
        raise RuntimeError(f"{type(module).__name__}: the BoostMVSNeRFs hot path runs on the GPU only (input is on "
                           f"{x.device}); there is no CPU fallback")
    return (not module.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
            and switches.get("BMV_CNN") != "torch")


# (BMV_FPN_FUSE / BMV_CONV0_FUSE / BMV_TOP_FUSE are read where FeatureNet's engine path runs: `switches.set()` /
# `override()` after import take effect on the next forward)

class _Packed:
    """Folded + packed weights of a module, rebuilt when any parameter / buffer changes (in-place updates
    bump `_version`; `.to()` / `.cuda()` replace storage and change `data_ptr`)."""

    def __init__(self):
        self.key = None
        self.blobs = None
        self.tensors = None

    def invalidate(self):
        self.key = self.tensors = None

    def get(self, module, build):
        if self.tensors is None:
            self.tensors = list(module.parameters()) + list(module.buffers())
        key = [t._version for t in self.tensors]
        key.append(self.tensors[0].data_ptr())
        key.append(self.tensors[-1].data_ptr())
        if key != self.key:
            self.tensors = list(module.parameters()) + list(module.buffers())    # buffers are replaced by .to()
            self.blobs, self.key = build(), key
        return self.blobs


def _pack_cbr(m):
    """_ConvBN -> (wpack, bias) with the batch norm folded in."""
    return convnet.pack_conv(*convnet.fold_bn(m.conv.weight, m.bn), stride=m.conv.stride[0])


def _bn_forward(bn, x, relu):
    """bn(x) [+ ReLU].  A plain nn.BatchNorm{2,3}d in training mode on the GPU runs on csrc/bn.hip (statistics pass +
    apply pass, ReLU fused; MIOpen's spatial batch norm costs ~73 us per call whatever the size); everything else --
    eval mode, SyncBatchNorm under DDP, BMV_BN=torch -- stays on torch."""
    if (type(bn) in (nn.BatchNorm2d, nn.BatchNorm3d) and bn.training and bn.track_running_stats and bn.affine
            and bn.momentum is not None and x.is_cuda and x.dtype == torch.float32
            and switches.get("BMV_BN") != "torch"):
        bn.num_batches_tracked.add_(1)
        return A.BatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)
    y = bn(x)
    return F.relu(y, inplace=True) if relu else y


class _ConvBN(nn.Module):
    """conv (no bias) -> batch norm -> ReLU; children named `conv` and `bn`."""

    def __init__(self, conv_cls, bn_cls, cin, cout, k, stride, pad):
        super().__init__()
        self.conv = conv_cls(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = bn_cls(cout)

    def forward(self, x):
        return _bn_forward(self.bn, self.conv(x), True)


def cbr2(cin, cout, k=3, stride=1, pad=1):
    return _ConvBN(Conv2d, nn.BatchNorm2d, cin, cout, k, stride, pad)


def cbr3(cin, cout, stride=1):
    return _ConvBN(Conv3d, nn.BatchNorm3d, cin, cout, 3, stride, 1)


class _Up3(nn.Sequential):
    """transposed conv -> batch norm (no ReLU); a Sequential so that the state-dict keys stay `convN.0.*`, `convN.1.*`"""

    def forward(self, x):
        return _bn_forward(self[1], self[0](x), False)


def up3(cin, cout):
    return _Up3(ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False), nn.BatchNorm3d(cout))


_S_KERNELS_MIN_PIXELS = 600_000     # views x H x W from which csrc/fpn_s.hip's kernels beat the fp32 ones (see engine_bottom_up)


class FeatureNet(nn.Module):
    """3 -> (32 ch @ 1/4, 16 ch @ 1/2, 8 ch @ 1) feature pyramid with top-down path."""

    def __init__(self):
        super().__init__()
        widths = (8, 16, 32)
        self.conv0 = nn.Sequential(cbr2(3, widths[0]), cbr2(widths[0], widths[0]))
        self.conv1 = nn.Sequential(cbr2(widths[0], widths[1], 5, 2, 2), cbr2(widths[1], widths[1]))
        self.conv2 = nn.Sequential(cbr2(widths[1], widths[2], 5, 2, 2), cbr2(widths[2], widths[2]))
        self.toplayer = Conv2d(32, 32, 1)
        self.lat1 = Conv2d(16, 32, 1)
        self.lat0 = Conv2d(8, 32, 1)
        self.smooth1 = Conv2d(32, 16, 3, padding=1)
        self.smooth0 = Conv2d(32, 8, 3, padding=1)
        self._packed = _Packed()
        self.pack_lookup = False     # engine path: emit the full-resolution map as the renderer's lookup records
        self.quad_out = True         # engine path: the two maps the plane sweeps read come out quad-planar (ops.QuadFeats)

    @staticmethod
    def _top_down(coarse, lateral):
        return F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True) + lateral

    def _apply(self, fn, *args, **kwargs):      # .to() / .cuda() / .float(): buffers are replaced, storage moves
        self._packed.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def _blobs(self):
        return self._packed.get(self, lambda: {
            **{f"conv{i}.{j}": _pack_cbr(getattr(self, f"conv{i}")[j]) for i in range(3) for j in range(2)},
            # first layer of the first block, folded, as the fused kernel's producer reads it ((8,3,3,3), (8))
            "conv0.0_raw": tuple(t.float().contiguous() for t in convnet.fold_bn(self.conv0[0].conv.weight, self.conv0[0].bn)),
            # the whole first block for csrc/fpn_s.hip's conv0_s_kernel (second layer on the bf16 matrix cores; BMV_CONV0_S)
            "conv0_s": convnet.pack_conv0_s(*convnet.fold_bn(self.conv0[0].conv.weight, self.conv0[0].bn),
                                            *convnet.fold_bn(self.conv0[1].conv.weight, self.conv0[1].bn)),
            # the encoder's 5x5 stride-2 / 3x3 layers for csrc/conv2d_s.hip (bf16 matrix cores; BMV_CONV2D_S)
            **{f"conv{i}.{j}_s": convnet.pack_conv2d_s(*convnet.fold_bn(getattr(self, f"conv{i}")[j].conv.weight,
                                                                         getattr(self, f"conv{i}")[j].bn))
               for i in (1, 2) for j in range(2)},
            "toplayer": convnet.pack_conv(self.toplayer.weight, self.toplayer.bias),
            "smooth1": convnet.pack_conv(self.smooth1.weight, self.smooth1.bias),
            "smooth0": convnet.pack_conv(self.smooth0.weight, self.smooth0.bias),
            # the same layer with its output channels in the order of the renderer's lookup records
            "smooth0_eo": convnet.pack_conv(self.smooth0.weight[list(convnet.LookupRecords.EVEN_ODD)],
                                            self.smooth0.bias[list(convnet.LookupRecords.EVEN_ODD)]),
            # smooth0 with lat0 folded in, split for the bf16 matrix cores (csrc/fpn_s.hip; BMV_FPN_S)
            "smooth0_s": convnet.pack_fpn_smooth_s(self.smooth0.weight, self.smooth0.bias, self.lat0.weight, self.lat0.bias),
            "smooth0_s_eo": convnet.pack_fpn_smooth_s(self.smooth0.weight, self.smooth0.bias, self.lat0.weight, self.lat0.bias,
                                                      order=convnet.LookupRecords.EVEN_ODD)})

    def engine_bottom_up(self, x):
        """Encoder + top layer: (c0, c1, p2, p2) with p2 a (N,32,H/4,W/4) view of the channel-last buffer the level-0
        sweep reads.  The coarsest map is all the level-0 cost volume needs, so a caller can start that cascade level
        while `engine_top_down` is still running."""
        P = self._blobs()
        # (the strip-walking bf16 kernels need ~1500 waves to fill the chip: 3 x 512 x 640 has 1881, 3 x 256 x 320 only 435
        # and runs the fp32 kernels faster -- 12.7 against 18.0 us, profiles/r6/fpn_s_rows.txt)
        big = x.shape[0] * x.shape[-2] * x.shape[-1] >= _S_KERNELS_MIN_PIXELS
        if switches.on("BMV_CONV0_FUSE") and switches.on("BMV_CONV0_S") and big:
            c0 = convnet.conv0_s(x, *P["conv0_s"])      # ... with the second layer on the bf16 matrix cores (csrc/fpn_s.hip)
        elif switches.on("BMV_CONV0_FUSE"):    # the 3-channel first layer is computed in the second layer's tile producer: one launch
            c0 = convnet.conv0_fused(x, *P["conv0.0_raw"], *P["conv0.1"], 8)
        else:
            c0 = convnet.conv_fwd(x, *P["conv0.0"], 8, 1, 3, relu=True)
            c0 = convnet.conv_fwd(c0, *P["conv0.1"], 8, 1, 3, relu=True)
        if switches.on("BMV_CONV2D_S") and big and x.shape[-1] % 4 == 0 and x.shape[-2] % 4 == 0:
            # conv1, conv2.0 on the bf16 matrix cores (csrc/conv2d_s.hip: weights stationary, independent strip-walking waves)
            if switches.on("BMV_CONV2D_S_REC"):
                # ... the maps BETWEEN them as split records (convnet.SplitRecords: split once by the producer's epilogue, staged
                # by LDS-DMA; bit-identical results); c1 also planar for the top-down step
                c1 = convnet.conv2d_s(c0, *P["conv1.0_s"], 16, 5, 2, relu=True, records=True)
                c1, c1r = convnet.conv2d_s(c1, *P["conv1.1_s"], 16, 3, 1, relu=True, records="both")
                c2 = convnet.conv2d_s(c1r, *P["conv2.0_s"], 32, 5, 2, relu=True)
            else:
                c1 = convnet.conv2d_s(c0, *P["conv1.0_s"], 16, 5, 2, relu=True)
                c1 = convnet.conv2d_s(c1, *P["conv1.1_s"], 16, 3, 1, relu=True)
                c2 = convnet.conv2d_s(c1, *P["conv2.0_s"], 32, 5, 2, relu=True)
        else:
            c1 = convnet.conv_fwd(c0, *P["conv1.0"], 16, 1, 5, 2, relu=True)
            c1 = convnet.conv_fwd(c1, *P["conv1.1"], 16, 1, 3, relu=True)
            c2 = convnet.conv_fwd(c1, *P["conv2.0"], 32, 1, 5, 2, relu=True)
        # the coarsest map is written once, channel-last (the level-0 sweep's layout); the top-down step reads it so
        quad = self.quad_out      # the plane sweep's quad-planar layout (inference default) or channel-last
        if switches.on("BMV_TOP_FUSE"):      # conv2.1 + toplayer: the 1x1 layer is a second stage of the 3x3 layer's workgroups
            p2 = convnet.conv_top(c2, *P["conv2.1"], *P["toplayer"], quad=quad)
        else:
            c2 = convnet.conv_fwd(c2, *P["conv2.1"], 32, 1, 3, relu=True)
            p2 = convnet.conv_fwd(c2, *P["toplayer"], 32, 1, 1, channels_last="quad" if quad else True)
        p2 = ops.QuadFeats(p2) if quad else p2.permute(0, 3, 1, 2)
        return c0, c1, p2, p2

    def engine_top_down(self, c0, c1, p2, rgb=None, defer_f0=False):
        """Top-down path + smoothing: (16 ch @ 1/2 channel-last view, 8 ch @ 1 planar).  With `rgb` (the source images
        (N,3,H,W)) the full-resolution map comes out as the fused renderer's lookup records instead
        (convnet.LookupRecords: feature channels + colours of a pixel in one 48-byte record)."""
        P = self._blobs()
        p1 = convnet.fpn_topdown(c1, p2, self.lat1.weight, self.lat1.bias)

        def full_resolution():
            fpn_s = (switches.on("BMV_FPN_S") and switches.on("BMV_FPN_FUSE") and c0.shape[1] == 8 and p1.shape[1] == 32
                     and c0.shape[0] * c0.shape[-2] * c0.shape[-1] >= _S_KERNELS_MIN_PIXELS)
            if fpn_s:       # one launch on the bf16 matrix cores, lat0 folded into smooth0's weights (csrc/fpn_s.hip)
                return convnet.fpn_smooth_s(c0, p1, *P["smooth0_s_eo" if rgb is not None else "smooth0_s"], rgb=rgb)
            if rgb is not None:
                return convnet.fpn_smooth(c0, p1, self.lat0.weight, self.lat0.bias, *P["smooth0_eo"], 8, rgb=rgb)
            if switches.on("BMV_FPN_FUSE"):
                # the full-resolution 32-channel map exists only between lat0 / upsample and smooth0: one launch, never written
                return convnet.fpn_smooth(c0, p1, self.lat0.weight, self.lat0.bias, *P["smooth0"], 8)
            p0 = convnet.fpn_topdown(c0, p1, self.lat0.weight, self.lat0.bias)
            return convnet.conv_fwd(p0, *P["smooth0"], 8, 1, 3)
        # (defer_f0: the full-resolution map -- only the renderer reads it -- is left to the caller, who may run it beside
        # the level-1 chain: `f0` comes back as the function that launches it)
        f0 = full_resolution if defer_f0 else full_resolution()
        # the level-1 sweep's source map LAST: it then sits in the L2s (rows band k in XCD k's, conv.hip's band map) when
        # that sweep starts instead of being pushed out by the 23 MB the full-resolution map writes
        f1 = convnet.conv_fwd(p1, *P["smooth1"], 16, 1, 3, channels_last="quad" if self.quad_out else True)
        return (ops.QuadFeats(f1) if self.quad_out else f1.permute(0, 3, 1, 2)), f0

    def _forward_engine(self, x):
        """Same graph, one launch per conv block.  The two maps the plane sweeps read come out quad-planar
        (ops.QuadFeats: `.to_nchw()` gives the planar tensor); with `quad_out = False` channel-last ((N,C,H,W) views of
        (N,H,W,C) buffers: `.contiguous()` gives the planar tensor)."""
        c0, c1, p2, p2_cl = self.engine_bottom_up(x)
        f1, f0 = self.engine_top_down(c0, c1, p2, rgb=x if self.pack_lookup else None)
        return p2_cl, f1, f0

    def forward(self, x):
        if _engine_ok(self, x):
            return self._forward_engine(x)
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        p2 = self.toplayer(c2)
        p1 = self._top_down(p2, self.lat1(c1))
        p0 = self._top_down(p1, self.lat0(c0))
        return p2, self.smooth1(p1), self.smooth0(p0)


class _CostReg(nn.Module):
    """3-D U-Net; `depth` encoder stages beyond the first (2 = MinCostRegNet, 3 = CostRegNet)."""

    def __init__(self, in_channels, depth):
        super().__init__()
        self.depth = depth
        self.conv0 = cbr3(in_channels, 8)
        self.conv1, self.conv2 = cbr3(8, 16, 2), cbr3(16, 16)
        self.conv3, self.conv4 = cbr3(16, 32, 2), cbr3(32, 32)
        if depth == 3:
            self.conv5, self.conv6 = cbr3(32, 64, 2), cbr3(64, 64)
            self.conv7 = up3(64, 32)
        self.conv9 = up3(32, 16)
        self.conv11 = up3(16, 8)
        self.depth_conv = nn.Sequential(Conv3d(8, 1, 3, padding=1, bias=False))
        self.feat_conv = nn.Sequential(Conv3d(8, 8, 3, padding=1, bias=False))
        self._packed = _Packed()
        self.volume_records = False   # engine path: emit the feature volume as the renderer's voxel records
        # first layer and heads on the bf16 matrix cores with split fp32 operands (csrc/conv_split.hip): 0 = fp32 MFMA
        # engine (default), "auto" = three pieces (fp32-equivalent) where faster stand-alone, 3 / 2 = all with 3 / 2 pieces
        self.split_bf16 = convnet.split_bf16_default()
        # first layer (32 | 16 -> 8) and heads (8 -> 8 + 1) on v_mfma_f32_4x4x1 (csrc/conv_c4.hip: every matrix row useful
        # for 8 output channels; the 16-row kernels of conv.hip reach 75 % / 56 %): same fp32 FMA chains per output
        self.conv_c4 = switches.on("BMV_CONV_C4")
        # ... and on the bf16 matrix cores with three-piece fp32 operands where the input arrives as quad records
        # (csrc/conv_c4s.hip: the sums of the fp32 kernel to fp32 rounding, 3.4 x the matrix rate)
        self.conv_c4s = switches.on("BMV_CONV_C4S")
        self.quad_volume = switches.on("BMV_QUAD_VOLUME")
        self.quad_s0 = switches.on("BMV_QUAD_S0")

    def takes_quad_volume(self):
        """True: hand forward() the cost volume as ops.QuadVolume (the sweep's quad-record output) -- the first layer runs
        on the 4-row-block kernel, which stages such an input with one 16-byte load per position."""
        return (self.quad_volume and self.conv_c4 and not self.training and not torch.is_grad_enabled()
                and switches.get("BMV_CNN") != "torch" and not convnet.split_parts(self.split_bf16, "conv0", self.conv0.conv.weight.shape[1]))

    def _apply(self, fn, *args, **kwargs):
        self._packed.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def prepack(self):
        """Folded + packed weights (cached).  Callers that run this module on several streams call it once on the
        stream they fork from: the pack kernels must not race with a side stream's first use of the blobs."""
        def build():
            P = {f"conv{i}": _pack_cbr(getattr(self, f"conv{i}")) for i in range(5 + 2 * (self.depth == 3))}
            for name in ("conv7", "conv9", "conv11")[3 - self.depth:]:
                up = getattr(self, name)
                P[name] = convnet.pack_convT(*convnet.fold_bn(up[0].weight, up[1], out_dim=1))
            # feat_conv (8 ch) and depth_conv (1 ch) read the same tensor: one 9-channel convolution
            heads = torch.cat([self.feat_conv[0].weight, self.depth_conv[0].weight], 0)
            P["heads"] = convnet.pack_conv(heads, None)
            # the same layer with its output channels in the order of the renderer's volume records
            P["heads_rec"] = convnet.pack_conv(heads[list(convnet.VolumeRecords.ORDER)], None)
            # first layer and heads on the 4 x 4 x 1 matrix blocks (csrc/conv_c4.hip; used when self.conv_c4)
            P["conv0_c4"] = convnet.pack_conv_c4(*convnet.fold_bn(self.conv0.conv.weight, self.conv0.bn))
            P["heads_c4"] = convnet.pack_conv_c4(heads, None)
            P["heads_rec_c4"] = convnet.pack_conv_c4(heads[list(convnet.VolumeRecords.ORDER)], None)
            P["conv11_c4"] = convnet.pack_convT_c4(*convnet.fold_bn(self.conv11[0].weight, self.conv11[1], out_dim=1))
            if self.conv0.conv.weight.shape[1] % 8 == 0:      # (csrc/conv_c4s.hip; used when self.conv_c4s)
                P["conv0_c4s"] = convnet.pack_conv_c4s(*convnet.fold_bn(self.conv0.conv.weight, self.conv0.bn))
            P["heads_c4s"] = convnet.pack_conv_c4s(heads, None)
            P["heads_rec_c4s"] = convnet.pack_conv_c4s(heads[list(convnet.VolumeRecords.ORDER)], None)
            for parts in (2, 3):     # (csrc/conv_split.hip; used when self.split_bf16 == parts; < 100 KB per regulariser)
                P[f"conv0_split{parts}"] = convnet.pack_conv_split(*convnet.fold_bn(self.conv0.conv.weight, self.conv0.bn), parts=parts)
                P[f"heads_split{parts}"] = convnet.pack_conv_split(heads, None, parts=parts)
                P[f"heads_rec_split{parts}"] = convnet.pack_conv_split(heads[list(convnet.VolumeRecords.ORDER)], None, parts=parts)
            return P
        return self._packed.get(self, build)

    def _forward_engine(self, x):
        P = self.prepack()
        ok4 = x.shape[-1] % 4 == 0
        split = convnet.split_parts(self.split_bf16, "conv0", x.shape[1]) if ok4 else 0
        if isinstance(x, ops.QuadVolume) and (split or not self.conv_c4):
            x = x.to_planar()
        if split:
            s0 = convnet.conv3d_split_fwd(x, *P[f"conv0_split{split}"], 8, relu=True)
        elif self.conv_c4 and self.conv_c4s and isinstance(x, ops.QuadVolume) and "conv0_c4s" in P:
            s0 = convnet.conv_c4s_fwd(x, *P["conv0_c4s"], 8, relu=True, quad_out=self.quad_volume and self.quad_s0)
        elif self.conv_c4:
            # (quad records out as well: the stride-2 layer and conv11's skip add stage them with 16-byte loads)
            s0 = convnet.conv_c4_fwd(x, *P["conv0_c4"], 8, relu=True, quad_out=self.quad_volume and self.quad_s0)
        else:
            s0 = convnet.conv_fwd(x, *P["conv0"], 8, 3, 3, relu=True)
        s1 = convnet.conv_fwd(convnet.conv_fwd(s0, *P["conv1"], 16, 3, 3, 2, relu=True), *P["conv2"], 16, 3, 3, relu=True)
        s2 = convnet.conv_fwd(convnet.conv_fwd(s1, *P["conv3"], 32, 3, 3, 2, relu=True), *P["conv4"], 32, 3, 3, relu=True)
        y = s2
        if self.depth == 3:
            t = convnet.conv_fwd(convnet.conv_fwd(s2, *P["conv5"], 64, 3, 3, 2, relu=True), *P["conv6"], 64, 3, 3, relu=True)
            y = convnet.convT3d_fwd(t, *P["conv7"], 32, skip=s2)
        y = convnet.convT3d_fwd(y, *P["conv9"], 16, skip=s1)
        split = convnet.split_parts(self.split_bf16, "heads", 8) if ok4 else 0
        if self.conv_c4:
            # (the heads run on the same 4-row-block kernel: they take conv11's result as quad records)
            y = convnet.convT_c4_fwd(y, *P["conv11_c4"], 8, skip=s0, quad_out=self.quad_volume and not split)
        else:
            y = convnet.convT3d_fwd(y, *P["conv11"], 8, skip=s0)
        if self.volume_records:      # the feature volume as the fused renderer's 32-byte voxel records
            if split:
                return convnet.conv3d_split_heads_records(y, *P[f"heads_rec_split{split}"])
            if self.conv_c4 and self.conv_c4s and isinstance(y, ops.QuadVolume):
                return convnet.conv_c4s_fwd(y, *P["heads_rec_c4s"], 9, records=True)
            if self.conv_c4:
                return convnet.conv_c4_fwd(y, *P["heads_rec_c4"], 9, records=True)
            return convnet.conv_heads_records(y, *P["heads_rec"])
        if split:
            heads = convnet.conv3d_split_fwd(y, *P[f"heads_split{split}"], 9)
            return heads[:, :8], heads[:, 8]
        if self.conv_c4 and self.conv_c4s and isinstance(y, ops.QuadVolume):
            return convnet.conv_c4s_fwd(y, *P["heads_c4s"], 9, split_heads=True)
        if self.conv_c4:
            heads = convnet.conv_c4_fwd(y, *P["heads_c4"], 9)
            return heads[:, :8], heads[:, 8]
        heads = convnet.conv_fwd(y, *P["heads"], 9, 3, 3)
        return heads[:, :8], heads[:, 8]

    def forward(self, x):
        if _engine_ok(self, x):
            return self._forward_engine(x)
        if isinstance(x, ops.QuadVolume):
            x = x.to_planar()
        s0 = self.conv0(x)
        s1 = self.conv2(self.conv1(s0))
        s2 = self.conv4(self.conv3(s1))
        y = s2
        if self.depth == 3:
            y = s2 + self.conv7(self.conv6(self.conv5(s2)))
        y = s1 + self.conv9(y)
        y = s0 + self.conv11(y)
        fw, dw = self.feat_conv[0].weight, self.depth_conv[0].weight
        if torch.is_grad_enabled() and fw.requires_grad and dw.requires_grad:
            # both heads read the same tensor: one 9-channel convolution (one forward, one data gradient and one
            # weight-gradient pass over the full-resolution volume instead of two); autograd splits the gradient
            heads = _Conv3dFn.apply(y, torch.cat([fw, dw], 0), 1)
            return heads[:, :8], heads[:, 8]
        return self.feat_conv(y), self.depth_conv(y).squeeze(1)


class MinCostRegNet(_CostReg):
    def __init__(self, in_channels):
        super().__init__(in_channels, 2)


class CostRegNet(_CostReg):
    def __init__(self, in_channels):
        super().__init__(in_channels, 3)
