"""2-D feature pyramid and 3-D cost regularisers of ENeRF.

These convolution stacks sit between the hot-path kernels (SURVEY.md section 8f:
"next" rows, not hot-path kernels); they stay torch modules and run on MIOpen.
Module/parameter names reproduce the reference's state-dict keys exactly
(lib/networks/enerf/feature_net.py:4-36, cost_reg_net.py:4-86, utils.py:10-33)
so `load_state_dict(ckpt['net'], strict=True)` accepts reference checkpoints.
"""
import torch.nn as nn
import torch.nn.functional as F

from .conv3d_wgrad import Conv3d, ConvTranspose3d   # MIOpen forward / data grad, slice-GEMM weight grad


class _ConvBN(nn.Module):
    """conv (no bias) -> batch norm -> ReLU; children named `conv` and `bn`."""

    def __init__(self, conv_cls, bn_cls, cin, cout, k, stride, pad):
        super().__init__()
        self.conv = conv_cls(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = bn_cls(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


def cbr2(cin, cout, k=3, stride=1, pad=1):
    return _ConvBN(nn.Conv2d, nn.BatchNorm2d, cin, cout, k, stride, pad)


def cbr3(cin, cout, stride=1):
    return _ConvBN(Conv3d, nn.BatchNorm3d, cin, cout, 3, stride, 1)


def up3(cin, cout):
    return nn.Sequential(ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm3d(cout))


class FeatureNet(nn.Module):
    """3 -> (32 ch @ 1/4, 16 ch @ 1/2, 8 ch @ 1) feature pyramid with top-down path."""

    def __init__(self):
        super().__init__()
        widths = (8, 16, 32)
        self.conv0 = nn.Sequential(cbr2(3, widths[0]), cbr2(widths[0], widths[0]))
        self.conv1 = nn.Sequential(cbr2(widths[0], widths[1], 5, 2, 2), cbr2(widths[1], widths[1]))
        self.conv2 = nn.Sequential(cbr2(widths[1], widths[2], 5, 2, 2), cbr2(widths[2], widths[2]))
        self.toplayer = nn.Conv2d(32, 32, 1)
        self.lat1 = nn.Conv2d(16, 32, 1)
        self.lat0 = nn.Conv2d(8, 32, 1)
        self.smooth1 = nn.Conv2d(32, 16, 3, padding=1)
        self.smooth0 = nn.Conv2d(32, 8, 3, padding=1)

    @staticmethod
    def _top_down(coarse, lateral):
        return F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True) + lateral

    def forward(self, x):
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

    def forward(self, x):
        s0 = self.conv0(x)
        s1 = self.conv2(self.conv1(s0))
        s2 = self.conv4(self.conv3(s1))
        y = s2
        if self.depth == 3:
            y = s2 + self.conv7(self.conv6(self.conv5(s2)))
        y = s1 + self.conv9(y)
        y = s0 + self.conv11(y)
        return self.feat_conv(y), self.depth_conv(y).squeeze(1)


class MinCostRegNet(_CostReg):
    def __init__(self, in_channels):
        super().__init__(in_channels, 2)


class CostRegNet(_CostReg):
    def __init__(self, in_channels):
        super().__init__(in_channels, 3)
