"""ctypes binding of libbmv.so (include/bmv.h).

There is NO fallback: if the HIP library is missing or a call fails, an
exception is raised.  The product path never routes through torch ops or the
CPU oracle for the functions this library implements.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMV_LIB_PATH") or os.path.join(_HERE, "libbmv.so")   # override: kernel experiments only

_lib = None

c_f = C.c_void_p   # device pointers travel as void*
c_i = C.c_int
c_l = C.c_long
c_fl = C.c_float


class NerfParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "view_fc_w", "view_fc_b", "global_fc_w", "global_fc_b", "agg_w_w", "agg_w_b", "fc_w", "fc_b",
        "lr0_w", "lr0_b", "sigma_w", "sigma_b", "color0_w", "color0_b", "color2_w", "color2_b")]


class RenderArgs(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("rays", "depth", "std", "near_far", "volume", "im_feat", "rgb_src",
                                           "src_exts", "src_ixts", "tar_ext", "blob")]
                + [(n, C.c_int) for n in ("B", "N", "S", "feat_ch", "Ns", "depth_inv", "hv", "wv", "Dv", "Hr", "Wr")]
                + [("render_scale", C.c_float)]
                + [(n, C.c_int) for n in ("rgb_affine", "white_bkgd", "mode", "ray_begin", "ray_end")]
                + [(n, C.c_void_p) for n in ("out0", "out1", "out2")]
                + [("view_ids", C.c_void_p), ("n_all", C.c_int), ("im_packed", C.c_void_p), ("vol_packed", C.c_int)])


class MvsMlpParams(C.Structure):
    _fields_ = ([("pts_w", C.c_void_p * 6), ("pts_b", C.c_void_p * 6)]
                + [(n, C.c_void_p) for n in ("bias_w", "bias_b", "views_w", "views_b", "feature_w", "feature_b",
                                             "alpha_w", "alpha_b", "rgb_w", "rgb_b")])


class MvsRenderArgs(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("rays", "volume", "src_inps", "src_exts", "src_ixts", "near_far", "blob")]
                + [(n, C.c_int) for n in ("N", "Ns", "S", "D", "hp", "wp", "H", "W", "pad", "ray_begin", "ray_end")]
                + [(n, C.c_void_p) for n in ("raw", "z_vals", "mask", "inputs86")])


# name -> argtypes (all return int unless noted); mirrors include/bmv.h one to one
SIGNATURES = {
    "bmv_proj_mats": [c_f, c_f, c_f, c_f, c_fl, c_fl, c_i, c_i, c_f, c_f],
    "bmv_depth_values_uniform": [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_frame_setup": [c_f, c_f, c_f, c_f, C.POINTER(C.c_float), C.POINTER(C.c_float), c_i, c_i, c_i, c_f, c_f, c_i, c_i,
                        c_i, c_i, c_f, c_f, c_f],
    "bmv_depth_values_cascade": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_homo_warp_fwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_sweep_variance_fwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_f],
    "bmv_sweep_variance_quad_fwd": [c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_f],
    "bmv_to_quad_planar": [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_sweep_variance_views_fwd": [c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_nchw_to_nhwc": [c_f, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_depth_regress_fwd": [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_build_rays": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_sample_along_depth": [c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_unpreprocess": [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_vox_feat": [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_img_feat": [c_f, c_f, c_f, c_f, c_f, c_fl, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_nerf_blob_size": [c_i],
    "bmv_nerf_pack_weights": [C.POINTER(NerfParams), c_i, c_f, c_f],
    "bmv_nerf_mlp_fwd": [c_f, c_f, c_f, c_i, c_i, c_l, c_f, c_f],
    "bmv_composite_fwd": [c_f, c_f, c_l, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_mask_viewport": [c_f, c_f, c_f, c_fl, c_fl, c_i, c_i, c_i, c_f, c_f],
    "bmv_ndc_coords": [c_f, c_f, c_f, c_fl, c_fl, c_i, c_i, c_f, c_f],
    "bmv_blend_fwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_render_rays_fwd": [C.POINTER(RenderArgs), c_f],
    "bmv_mvs_proj_mats": [c_f, c_f, c_i, c_i, c_f, c_f],
    "bmv_resize_bilinear": [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_mvs_sweep_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_mvs_sweep_cl_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_mvs_mlp_blob_size": [],
    "bmv_mvs_mlp_pack_weights": [C.POINTER(MvsMlpParams), c_f, c_f],
    "bmv_mvs_mlp_fwd": [c_f, c_f, c_l, c_f, c_f],
    "bmv_mvs_mlp_train_act_floats": [c_l],
    "bmv_mvs_mlp_train_scratch_floats": [],
    "bmv_mvs_mlp_train_fwd": [c_f, C.POINTER(MvsMlpParams), c_l, c_f, c_f, c_f, c_f],
    "bmv_mvs_mlp_train_bwd": [C.POINTER(MvsMlpParams), c_f, c_f, c_f, c_f, c_l, c_f, C.POINTER(MvsMlpParams), c_f],
    "bmv_mvs_render_fwd": [C.POINTER(MvsRenderArgs), c_f],
    "bmv_mvs_march_mask": [c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_fl, c_f, c_f, c_f],
    "bmv_conv_wpack_floats": [c_i, c_i, c_i, c_i, c_i],
    "bmv_conv_pairs_rows": [c_i, c_i, c_i, c_i],
    "bmv_conv_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f],
    "bmv_conv3d_transpose_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "bmv_fpn_topdown_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "bmv_conv_c4_wpack_floats": [c_i, c_i, c_i],
    "bmv_conv_c4_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_i, c_f],
    "bmv_conv0_s_wsplit_ints": [],
    "bmv_conv2d_s_wsplit_ints": [c_i, c_i, c_i, c_i],
    "bmv_conv2d_s_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "bmv_conv0_s_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_f],
    "bmv_fpn_smooth_s_wsplit_ints": [],
    "bmv_fpn_smooth_s_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_f],
    "bmv_conv_c4s_wsplit_ints": [c_i, c_i],
    "bmv_conv_c4s_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f],
    "bmv_conv3d_transpose_c4_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f],
    "bmv_conv3d_split_wsplit_ints": [c_i, c_i],
    "bmv_conv3d_split_fwd": [c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, C.c_float, c_f],
    "bmv_conv3d_split_heads_fwd": [c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f],
    "bmv_make_rays": [c_f, c_f, c_i, c_i, c_i, C.c_double, c_f, c_f],
    "bmv_composite_bwd": [c_f, c_f, c_f, c_f, c_l, c_i, c_f, c_f],
    "bmv_blend_bwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_vox_feat_bwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_img_feat_bwd": [c_f, c_f, c_f, c_f, c_f, c_fl, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_sample_along_depth_bwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_build_rays_bwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_depth_regress_bwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_depth_values_cascade_bwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_sweep_variance_bwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_sweep_variance_bwd_cl": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_fixed_workspace": [c_l],
    "bmv_vox_feat_bwd_fixed": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_img_feat_bwd_fixed": [c_f, c_f, c_f, c_f, c_f, c_fl, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_build_rays_bwd_fixed": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_depth_values_cascade_bwd_fixed": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_sweep_variance_bwd_fixed": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f],
    "bmv_mvs_sweep_bwd_fixed": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_mvs_vol_feat_bwd_fixed": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_nerf_bwd_blob_size": [c_i],
    "bmv_nerf_bwd_rows": [c_i, c_i, C.POINTER(C.c_int)],
    "bmv_nerf_pack_bwd_weights": [C.POINTER(NerfParams), c_i, c_f, c_f],
    "bmv_nerf_bwd_workspace": [c_i, c_i, c_l],
    "bmv_nerf_mlp_bwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_l, c_f, c_f, c_f, C.POINTER(NerfParams), c_f],
    "bmv_conv_wgrad_workspace": [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i],
    "bmv_conv_wgrad": [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "bmv_mvs_sweep_bwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_mvs_vol_feat_bwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_conv_pack_weights": [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    "bmv_bn_chunks": [c_i, c_l],
    "bmv_bn_train_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_l, c_fl, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f],
    "bmv_bn_train_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_l, c_fl, c_f, c_f, c_f, c_f, c_f],
    "bmv_conv_top_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_i, c_f],
    "bmv_conv0_fused_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_fl, c_f],
    "bmv_conv_heads_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f],
    "bmv_fpn_smooth_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "bmv_event_create": [C.POINTER(C.c_void_p)],
    "bmv_event_destroy": [C.c_void_p],
    "bmv_event_record": [C.c_void_p, C.c_void_p],
    "bmv_event_elapsed_us": [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)],
    "bmv_bind_next_launch": [C.c_void_p, C.c_void_p],
    "bmv_launch_events_pending": [],
    "bmv_defer_pointer": [C.c_void_p, C.c_void_p, c_i],
    "bmv_deferred_pending": [],
    "bmv_ptr_table_set": [C.c_void_p, c_i, C.POINTER(C.c_int), C.POINTER(C.c_void_p), c_f],
    "bmv_frame_feed": [C.c_void_p, c_i, C.POINTER(C.c_int), C.POINTER(C.c_void_p), c_i, C.POINTER(C.c_void_p),
                       C.POINTER(C.c_void_p), C.POINTER(C.c_int), c_f],
    "bmv_frame_feed_ring": [C.c_void_p, C.c_void_p, C.c_void_p, c_i, c_f],
    "bmv_frame_feed_msg_bytes": [],
    "bmv_copy_to_slot": [c_f, C.c_void_p, c_i, c_l, c_f],
    "bmv_copy_to_slots": [c_i, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_long), c_f],
    "bmv_version": [],
    "bmv_render_pc_check": [c_i],
    "bmv_debug_render_pc_inject": [c_i],
    "bmv_tuning_set": [C.c_char_p, c_i],
    "bmv_tuning_clear": [C.c_char_p],
    "bmv_tuning_get": [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "bmv_tuning_name": [c_i],          # returns const char*
    "bmv_tuning_doc": [c_i],           # returns const char*
}


def load():
    """Load libbmv.so once; raises ImportError (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if os.environ.get("BMV_AUTOBUILD", "0") == "1":
            from . import build
            build.build(verbose=False)
        else:
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP kernels first "
                "(python -m boostmvsnerfs_amd.build, or __graft_entry__.build()). "
                "There is no CPU / torch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so is stale: loud by design
        fn.argtypes = args
        fn.restype = C.c_int
    lib.bmv_nerf_bwd_workspace.restype = C.c_long
    lib.bmv_fixed_workspace.restype = C.c_long
    lib.bmv_mvs_mlp_train_act_floats.restype = C.c_long
    lib.bmv_mvs_mlp_train_scratch_floats.restype = C.c_long
    lib.bmv_conv_wgrad_workspace.restype = C.c_long
    lib.bmv_last_error.argtypes = []
    lib.bmv_last_error.restype = C.c_char_p
    lib.bmv_tuning_name.restype = C.c_char_p
    lib.bmv_tuning_doc.restype = C.c_char_p
    _lib = lib
    # the launchers' tuning switches are explicit library state (include/bmv.h); environment variables of the same names
    # are applied HERE, once, in the open -- the C side never reads the environment
    for name in tuning_names():
        if name in os.environ:
            raw = os.environ[name].strip()
            try:
                value = int(raw, 0)
            except ValueError:
                try:
                    value = int(raw)          # "08" and the like, as the C side's atoi read them before round 4
                except ValueError:
                    raise ValueError(f"environment variable {name}={os.environ[name]!r}: expected an integer "
                                     "(a tuning switch of libbmv, include/bmv.h bmv_tuning_set)") from None
            set_tuning(name, value)
    for name in REMOVED_SWITCHES:
        if name in os.environ:
            import warnings
            warnings.warn(f"{name} is set but no longer read: its kernels were removed in round 4 (csrc/sweep_quad.hip is "
                          "the inference sweep; see bmv_tuning_name for the current switches)", stacklevel=2)
    return lib


REMOVED_SWITCHES = ("BMV_SWEEP_WIN", "BMV_SWEEP_ZP", "BMV_SWEEP_SPLIT", "BMV_RING_DEFS")


def tuning_names():
    lib = load()
    names, i = [], 0
    while True:
        n = lib.bmv_tuning_name(i)
        if n is None:
            return names
        names.append(n.decode())
        i += 1


def set_tuning(name, value):
    """Set (value = int) or clear (value = None) a launcher tuning switch (include/bmv.h: bmv_tuning_set)."""
    lib = load()
    if value is None:
        check(lib.bmv_tuning_clear(name.encode()), "tuning_clear")
    else:
        check(lib.bmv_tuning_set(name.encode(), int(value)), "tuning_set")


def get_tuning(name):
    """The switch's value, or None when it is not set."""
    v, is_set = C.c_int(0), C.c_int(0)
    check(load().bmv_tuning_get(name.encode(), C.byref(v), C.byref(is_set)), "tuning_get")
    return v.value if is_set.value else None


# Tensors whose addresses were taken for the launch being assembled.  `dptr(x.contiguous())` drops the last reference
# to the copy as soon as the address is taken, and the caching allocator hands the very block to the NEXT argument's
# copy of the same call -- whose copy kernel then overwrites the first argument before the launch reads it (round 4:
# two non-contiguous arguments in one call are all it takes).  Every tensor handed to dptr() is therefore kept
# referenced until the launch has been enqueued: check() -- called after every entry point -- lets go.
# Per THREAD, like the C side's deferral list: a check() on another thread must not let go of the tensors of a launch
# this thread is still assembling.
import threading  # noqa: E402

_tls = threading.local()


def _held_list():
    h = getattr(_tls, "held", None)
    if h is None:
        h = _tls.held = []
    return h


def check(rc, what=""):
    _held_list().clear()
    if rc != 0:
        msg = load().bmv_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libbmv {what} failed (code {rc}): {msg}")


def dptr(t, name="tensor", dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor of `dtype` (fp32 unless stated; validated)."""
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the BoostMVSNeRFs hot path runs on the GPU only (tensor is on {t.device}); "
                           "there is no CPU fallback")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    _held_list().append(t)
    return C.c_void_p(t.data_ptr())


def stream():
    """hipStream_t of torch's current stream on the current device.  Called once per launch: the raw-handle getters
    cost ~0.5 us where `torch.cuda.current_stream().cuda_stream` costs ~8 us (0.3 ms per eager 512x640 frame)."""
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:      # a torch build without the private getters
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
