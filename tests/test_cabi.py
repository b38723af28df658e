"""The C-ABI library loads and exports every symbol include/bmv.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "bmv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmv_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from boostmvsnerfs_amd import build
    path = build.build(verbose=False)          # no-op when the in-tree .so is current
    return ctypes.CDLL(path)


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ("bmv_sweep_variance_fwd", "bmv_depth_regress_fwd", "bmv_render_rays_fwd", "bmv_blend_fwd",
                 "bmv_nerf_mlp_fwd", "bmv_composite_fwd"):
        assert must in syms


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/bmv.h but not exported by libbmv.so: {missing}"


def test_binding_matches_header(lib):
    from boostmvsnerfs_amd import _lib
    declared = set(declared_symbols()) - {"bmv_last_error"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_host_only_entry_points(lib):
    lib.bmv_version.restype = ctypes.c_int
    assert lib.bmv_version() >= 1
    lib.bmv_nerf_blob_size.restype = ctypes.c_int
    n8, n32 = lib.bmv_nerf_blob_size(8), lib.bmv_nerf_blob_size(32)
    assert 10000 < n8 < n32 < 40000 and n8 % 4 == 0 and n32 % 4 == 0
    assert lib.bmv_nerf_blob_size(7) < 0
    lib.bmv_last_error.restype = ctypes.c_char_p
    assert b"unsupported" in lib.bmv_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from boostmvsnerfs_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.depth_regress(torch.zeros(1, 4, 2, 2), torch.ones(1, 4, 2, 2), True)


def test_cnn_modules_refuse_cpu_tensors():
    import torch
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    set_cfg(make_cfg("enerf_eval"))
    from boostmvsnerfs_amd.networks.enerf.cnn import FeatureNet
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FeatureNet().eval()(torch.zeros(1, 3, 32, 32))
