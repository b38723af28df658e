"""Import the reference (read-only, /root/reference) on CPU in THIS container.

Used only by tests/golden/make_golden.py to generate golden vectors; nothing
here runs on the GPU box (where /root/reference does not exist) and no
reference source is copied: the four modules the reference needs but this image
lacks are replaced by in-memory stand-ins:

  kornia.utils.create_meshgrid  -> (1,h,w,2) grid, [...,0]=x, [...,1]=y
  inplace_abn.InPlaceABN        -> batch norm + leaky_relu(0.01)
  cv2, torchvision.transforms   -> empty modules (import-time only)
"""
import os
import sys
import tempfile
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("BMV_REFERENCE_ROOT", "/root/reference")


def _install_stubs():
    kornia = types.ModuleType("kornia")
    kutils = types.ModuleType("kornia.utils")

    def create_meshgrid(height, width, normalized_coordinates=True, device="cpu", dtype=torch.float32):
        xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
        ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
        if normalized_coordinates:
            xs = (xs / (width - 1) - 0.5) * 2
            ys = (ys / (height - 1) - 0.5) * 2
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack([gx, gy], -1)[None]

    kutils.create_meshgrid = create_meshgrid
    kornia.utils = kutils
    sys.modules["kornia"] = kornia
    sys.modules["kornia.utils"] = kutils

    iabn = types.ModuleType("inplace_abn")

    class InPlaceABN(nn.Module):
        def __init__(self, num_features, **kw):
            super().__init__()
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
            self.register_buffer("running_mean", torch.zeros(num_features))
            self.register_buffer("running_var", torch.ones(num_features))
            self.eps, self.momentum = 1e-5, 0.1

        def forward(self, x):
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                             self.training, self.momentum, self.eps)
            return F.leaky_relu(y, 0.01)

    iabn.InPlaceABN = InPlaceABN
    sys.modules["inplace_abn"] = iabn

    cv2 = types.ModuleType("cv2")
    cv2.COLORMAP_JET = 2
    cv2.setNumThreads = lambda n: None
    sys.modules["cv2"] = cv2

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.ToTensor = type("ToTensor", (), {})
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt

    # the MVSNeRF Embedder calls .cuda() in its constructor (mvsnerf/network.py:44)
    torch.Tensor.cuda = lambda self, *a, **k: self


def load_reference(cfg_file, opts=()):
    """Returns the reference's `cfg` after importing lib.config with cfg_file."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    os.environ.setdefault("workspace", tempfile.mkdtemp(prefix="bmv_ws_"))
    os.chdir(REFERENCE_ROOT)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.argv = ["run.py", "--type", "evaluate", "--cfg_file", cfg_file, *opts]
    from lib.config import cfg  # noqa: E402  (argparse runs at import)
    return cfg
