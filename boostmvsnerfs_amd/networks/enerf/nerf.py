"""ENeRF's per-sample MLP (lib/networks/enerf/nerf.py:5-89) as a parameter holder
whose forward runs the MFMA kernels (csrc/mlp.hpp, mlp_bwd.hip) for S in {2, 3, 4} source
views per cost volume (the reference's Agg / NeRF are S-agnostic; ENeRF pre-trains with
train_input_views [2, 3, 4], configs/exps/pretrain/enerf/dtu_pretrain.yaml:22-23): no torch
math on this path.

State-dict keys equal the reference's: agg.{view_fc,global_fc,agg_w_fc,fc}.0,
lr0.0, sigma.0, color.{0,2} (weight, bias each).
"""
import torch
import torch.nn as nn

from ... import ops
from ...config import cfg


def _kaiming(linear):
    nn.init.kaiming_normal_(linear.weight.data)      # weights_init, nerf.py:130-134
    nn.init.zeros_(linear.bias.data)
    return linear


def _layer(cin, cout, act):
    return nn.Sequential(_kaiming(nn.Linear(cin, cout)), act)


class Agg(nn.Module):
    def __init__(self, feat_ch):
        super().__init__()
        self.feat_ch = feat_ch
        if not cfg.enerf.viewdir_agg:
            raise NotImplementedError("the HIP MLP implements viewdir_agg=True (every shipped config)")
        self.view_fc = _layer(4, feat_ch, nn.ReLU())
        self.global_fc = _layer(feat_ch * 3, 32, nn.ReLU())
        self.agg_w_fc = _layer(32, 1, nn.ReLU())
        self.fc = _layer(32, 16, nn.ReLU())


class NeRF(nn.Module):
    def __init__(self, hid_n=64, feat_ch=16 + 3):
        super().__init__()
        if hid_n != 64:
            raise NotImplementedError("MFMA MLP is laid out for hid_n=64")
        self.hid_n = hid_n
        self.feat_ch = feat_ch
        self.agg = Agg(feat_ch)
        self.lr0 = _layer(8 + 16, hid_n, nn.ReLU())
        self.lrs = nn.ModuleList([])
        self.sigma = _layer(hid_n, 1, nn.Softplus())
        self.color = nn.Sequential(_kaiming(nn.Linear(64 + 24 + feat_ch + 4, hid_n)), nn.ReLU(),
                                   _kaiming(nn.Linear(hid_n, 1)), nn.ReLU())
        self._blob = None
        self._blob_key = None

    def _linears(self):
        return (self.agg.view_fc[0], self.agg.global_fc[0], self.agg.agg_w_fc[0], self.agg.fc[0], self.lr0[0],
                self.sigma[0], self.color[0], self.color[2])

    def packed_weights(self):
        """MFMA-ordered weight blob; re-packed (one tiny kernel) whenever a parameter changed."""
        tensors = [t for lin in self._linears() for t in (lin.weight, lin.bias)]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._blob_key:
            self._blob = ops.nerf_pack_weights(tensors, self.feat_ch - 3, out=None)
            self._blob_key = key
        return self._blob

    def forward(self, vox_feat, img_feat_rgb_dir):
        """vox_feat (B,P,8), img_feat_rgb_dir (B,P,S,feat_ch+4) -> (B,P,4) = [rgb, sigma]; S in {2, 3, 4}."""
        if torch.is_grad_enabled():
            from ...autograd import NerfMLP
            params = [t for lin in self._linears() for t in (lin.weight, lin.bias)]
            return NerfMLP.apply(vox_feat, img_feat_rgb_dir, self.feat_ch - 3, *params)
        return ops.nerf_mlp(vox_feat, img_feat_rgb_dir, self.packed_weights(), self.feat_ch - 3)
