"""The convolution modules of the CNN stacks under autograd: nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d subclasses whose
forward, data gradients and WEIGHT gradients run on the package's own kernels (csrc/conv.hip, csrc/conv_wgrad.hip)
instead of MIOpen's solvers.

Measured on MI355X / ROCm 7.2 (scripts/probe_conv3d.py): MIOpen picks `naive_conv_*_wrw_ncdhw` or a 40 ms CK
batched-GEMM for every 3x3x3 fp32 layer of the cost regularisers (308 ms per CostRegNet backward, >90 % of a fine-tune
step; its exhaustive find mode takes >15 min).  The weight gradient of a k^3 convolution is k^3 products over the voxel
dimension,
    dW[:, :, kd, kh, kw] = dY (Co x P) @ X_shift(kd,kh,kw)^T (P x Ci),
which bmv_conv_wgrad computes with the voxel index as the MFMA k dimension (27 accumulator blocks per wave, no im2col
copies; the first version of this file made 27 strided copies and one tall-skinny rocBLAS GEMM per layer: 4 ms of
copies + ~5 ms of GEMMs per 512x640 step).  The forward runs on the inference engine with the weights repacked on the
device, the data gradients are convolutions with transformed filters on the same engine (stride-2 5x5: one 3x3
convolution over the four input parities + a pixel shuffle).
SURVEY.md section 8(f) ranks 1-2; module and parameter names are unchanged (subclasses of nn.Conv2d / nn.Conv3d /
nn.ConvTranspose3d).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import convnet, switches


def _engine_forward(x):
    """The training FORWARD of a convolution runs on the package's engine (csrc/conv.hip, weights repacked on the
    device every step: one launch) unless BMV_TRAIN_CONV=torch; the backward never does (MIOpen data gradients, own
    weight-gradient kernel for 3-D)."""
    return x.is_cuda and x.dtype == torch.float32 and switches.get("BMV_TRAIN_CONV") != "torch"


def _wgrad(big, small, stride, kd=3, k=3):
    """big: (B, Cb, [Db,] Hb, Wb) already zero-padded; small: (B, Cs, [Ds,] Hs, Ws).
    Returns G (Cs, Cb, [kd,] k, k) with G[s, b, tap] = sum_n sum_p small[n, s, p] * big[n, b, stride*p + tap]."""
    from ... import ops
    return ops.conv_wgrad(big, small, stride, kd, k)


class _Conv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride):
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        if _engine_forward(x):
            return convnet.conv_fwd(x, *convnet.pack_conv_dev(w, None, stride), w.shape[0], 3, 3, stride)
        return F.conv3d(x, w, None, stride, 1)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        s = ctx.stride
        gx = gw = None
        if ctx.needs_input_grad[0]:
            even = all(v % 2 == 0 for v in x.shape[2:])
            if _engine_forward(gy) and s == 1:
                # the adjoint of a stride-1 convolution is the convolution with the flipped, transposed filter
                gx = convnet.conv_fwd(gy, *convnet.pack_conv_dev(w, None, 1, transposed=True, flip=True), w.shape[1], 3, 3, 1)
            elif _engine_forward(gy) and s == 2 and even:
                # ... of a stride-2 one (even input sizes) the transposed convolution with the same filter
                gx = convnet.convT3d_fwd(gy, *convnet.pack_conv_dev(w, None, transposed=True, for_convT=True), w.shape[1])
            else:
                gx = torch.ops.aten.convolution_backward(gy, x, w, None, [s] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            gw = _wgrad(F.pad(x, (1, 2 if s == 2 else 1, 1, 1, 1, 1)), gy, s)          # (Co, Ci, 3,3,3)
        return gx, gw, None


class _ConvT3dFn(torch.autograd.Function):
    """ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1, bias=False)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        if _engine_forward(x):
            return convnet.convT3d_fwd(x, *convnet.pack_conv_dev(w, None, transposed=True, for_convT=True), w.shape[1])
        return F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:                                 # adjoint of the transposed conv: the stride-2 conv
            if _engine_forward(gy):
                gx = convnet.conv_fwd(gy, *convnet.pack_conv_dev(w, None, 2), w.shape[0], 3, 3, 2)
            else:
                gx = F.conv3d(gy, w, None, 2, 1)
        if ctx.needs_input_grad[1]:
            # y[o] += x[i] * w[k] with o = 2 i - 1 + k  ->  dW[ci, co, k] = sum_i x[ci, i] * dY[co, 2 i - 1 + k]
            gw = _wgrad(F.pad(gy, (1, 1, 1, 1, 1, 1)), x, 2)                           # (Ci, Co, 3,3,3)
        return gx, gw


_PARITY_TAPS = {}


def _dgrad_5x5_stride2(gy, w):
    """Data gradient of Conv2d(k=5, stride=2, padding=2) on even input sizes as ONE stride-1 3x3 convolution of gy on
    the engine + a pixel shuffle.  y[o] = sum_k w[k] x[2o + k - 2], so dx[2j + r] = sum_t w[2t + r] gy[j + 1 - t]
    (r = 0: t in 0..2, r = 1: t in 0..1): every input parity (ry, rx) is a <= 3x3-tap correlation of gy with the
    sub-filter g_r[u] = w[2 (2 - u) + r] (tap 5 = 0).  The four parities are 4 Cin output channels of one convolution
    in (ci, ry, rx) order -- exactly pixel_shuffle's layout.  36 tap products per input pixel instead of 25."""
    Co, Ci = w.shape[:2]
    wp = F.pad(w.detach(), (0, 1, 0, 1))                                     # (Co,Ci,6,6), index 5 = 0
    idx = _PARITY_TAPS.get(w.device)
    if idx is None:                                                          # (host -> device copy: once, never inside a capture)
        idx = _PARITY_TAPS[w.device] = torch.tensor([[4, 2, 0], [5, 3, 1]], device=w.device)
    g = wp[:, :, idx][..., idx]                                              # (Co,Ci,ry,uy,rx,ux)
    g = g.permute(1, 2, 4, 0, 3, 5).reshape(4 * Ci, Co, 3, 3).contiguous()   # (ci,ry,rx | co | uy,ux)
    planes = convnet.conv_fwd(gy, *convnet.pack_conv_dev(g, None, 1), 4 * Ci, 1, 3, 1)
    return F.pixel_shuffle(planes, 2)


class _Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (k in {1, 3, 5}, zero padding k // 2): forward, data gradients and weight gradients on the engine
    (odd input sizes under stride 2 fall back to aten)."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, b is not None)
        return convnet.conv_fwd(x, *convnet.pack_conv_dev(w, b, stride), w.shape[0], 1, w.shape[-1], stride)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        s, has_b = ctx.cfg
        p = w.shape[-1] // 2
        gy = gy.contiguous()
        k = w.shape[-1]
        want_x, eng = ctx.needs_input_grad[0], _engine_forward(gy)
        # stride 1: the convolution with the flipped, transposed filter; stride-2 5x5: four parity sub-filters
        s2 = s == 2 and k == 5 and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0 and switches.on("BMV_TRAIN_DGRAD5")
        gx_engine = want_x and (s == 1 or s2) and eng
        gw_engine = ctx.needs_input_grad[1] and eng                # weight gradient: own MFMA kernel over (batch, pixel)
        gx = gw = gb = None
        if (want_x and not gx_engine) or (ctx.needs_input_grad[1] and not gw_engine):
            gx, gw, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1,
                                                            [want_x and not gx_engine, ctx.needs_input_grad[1] and not gw_engine, False])
        if gx_engine and s == 1:
            gx = convnet.conv_fwd(gy, *convnet.pack_conv_dev(w, None, 1, transposed=True, flip=True), w.shape[1], 1, k, 1)
        elif gx_engine:
            gx = _dgrad_5x5_stride2(gy, w)
        if gw_engine:
            gw = _wgrad(F.pad(x, (p, p + (1 if s == 2 else 0), p, p)) if p else x, gy, s, 1, k)
        if has_b:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb, None


class Conv2d(nn.Conv2d):
    def forward(self, x):
        k = self.kernel_size[0]
        if (torch.is_grad_enabled() and _engine_forward(x) and self.kernel_size == (k, k) and (k, self.stride) in
                ((1, (1, 1)), (3, (1, 1)), (5, (2, 2))) and self.padding == (k // 2, k // 2) and self.dilation == (1, 1)
                and self.groups == 1 and self.padding_mode == "zeros"):
            return _Conv2dFn.apply(x, self.weight, self.bias, self.stride[0])
        return super().forward(x)


class Conv3d(nn.Conv3d):
    def forward(self, x):
        if (torch.is_grad_enabled() and self.weight.requires_grad and self.bias is None and self.kernel_size == (3, 3, 3)
                and self.padding == (1, 1, 1) and self.dilation == (1, 1, 1) and self.groups == 1
                and self.stride[0] == self.stride[1] == self.stride[2]):
            return _Conv3dFn.apply(x, self.weight, self.stride[0])
        return super().forward(x)


class ConvTranspose3d(nn.ConvTranspose3d):
    def forward(self, x, output_size=None):
        if (torch.is_grad_enabled() and self.weight.requires_grad and self.bias is None and output_size is None
                and self.kernel_size == (3, 3, 3) and self.stride == (2, 2, 2) and self.padding == (1, 1, 1)
                and self.output_padding == (1, 1, 1) and self.groups == 1):
            return _ConvT3dFn.apply(x, self.weight)
        return super().forward(x, output_size)
