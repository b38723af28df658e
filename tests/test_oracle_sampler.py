"""The oracle's numpy restatement of grid_sampler (oracle/enerf.py: grid_sample_2d_np / grid_sample_3d_np) against
ATen's, for the padding modes and align_corners setting the reference calls it with -- including coordinates far
outside the image, exactly on the border and on integer texels.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import enerf as O


def _grid(rng, n, dims):
    g = rng.uniform(-1.6, 1.6, (n, dims))
    g[:8] = rng.choice([-1.0, 1.0, 0.0], (8, dims))       # borders and the centre texel
    g[8:16] = np.round(g[8:16] * 4) / 4                   # integer texels of a 9-wide axis
    g[16:20] *= 50                                        # far outside
    return g


@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_sampler_2d_matches_aten(pad):
    rng = np.random.default_rng(0)
    img = rng.standard_normal((5, 9, 13))
    g = _grid(rng, 400, 2)
    want = F.grid_sample(torch.from_numpy(img)[None], torch.from_numpy(g)[None, None], mode="bilinear",
                         padding_mode=pad, align_corners=True)[0, :, 0].numpy()
    got = O.grid_sample_2d_np(img, g, pad)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_sampler_3d_matches_aten(pad):
    rng = np.random.default_rng(1)
    vol = rng.standard_normal((4, 6, 9, 7))
    g = _grid(rng, 400, 3)
    want = F.grid_sample(torch.from_numpy(vol)[None], torch.from_numpy(g)[None, None, None], mode="bilinear",
                         padding_mode=pad, align_corners=True)[0, :, 0, 0].numpy()
    got = O.grid_sample_3d_np(vol, g, pad)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_oracle_lookups_agree_with_the_numpy_sampler():
    """vox_lookup (a9) and the warp (a3) route through ATen; the same numbers come out of the numpy statement."""
    torch.manual_seed(0)
    vol = torch.randn(1, 8, 4, 6, 5, dtype=torch.float64)
    uvd = torch.rand(1, 50, 3, dtype=torch.float64) * 1.2 - 0.1
    want = O.vox_lookup(uvd, vol)[0].numpy().T
    got = O.grid_sample_3d_np(vol[0].numpy(), uvd[0].numpy() * 2 - 1, "zeros")
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
