"""Tensor-level wrappers over the C ABI (one function per entry point of
include/bmv.h).  Inputs must be CUDA fp32 tensors; outputs are allocated with
torch (device memory + stream plumbing only) and filled by the HIP kernels.
Everything runs on torch's current stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ktimer, switches
from ._lib import dptr, stream

BMV_ERR_UNSUPPORTED = -3      # include/bmv.h


def _c(t):
    """`t`, contiguous.  (A copy made here outlives the launch it is made for: `_lib.dptr` keeps every tensor it is
    given referenced until `_lib.check` -- see there.)"""
    return t if t.is_contiguous() else t.contiguous()


def proj_mats(src_exts, src_ixts, tar_ext, tar_ixt, src_scale, tar_scale):
    B, S = src_exts.shape[:2]
    out = torch.empty(B, S, 3, 4, device=src_exts.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_proj_mats(dptr(_c(src_exts), "src_exts"), dptr(_c(src_ixts), "src_ixts"),
                                 dptr(_c(tar_ext), "tar_ext"), dptr(_c(tar_ixt), "tar_ixt"),
                                 float(src_scale), float(tar_scale), B, S, dptr(out), stream()), "proj_mats")
    return out


def scaled_size(H, W, scale):
    """(h, w) of a level rendered at `scale` -- ONE definition for the ray grid, the renderer's Hr x Wr and the loss
    side.  The reference computes it twice: the network with int(H * scale) (lib/networks/enerf/network.py:26-27), the
    dataset with cv2.resize's rounding (lib/datasets/enerf_utils.py:62-66); they agree whenever H * scale is integral
    (true for every shipped size), and a configuration where they would not is refused here instead of rendering a
    ray grid the network reshapes to a different size."""
    if scale == 1.0:
        return int(H), int(W)
    h, w = int(H * scale), int(W * scale)
    if (h, w) != (int(round(H * scale)), int(round(W * scale))):
        raise ValueError(f"render scale {scale} of a {H}x{W} target: int() and round() disagree on the scaled size "
                         f"({h}x{w} vs {int(round(H * scale))}x{int(round(W * scale))}); the reference's network and dataset would too")
    return h, w


def make_rays(tar_ext, tar_ixt, H, W, scale=1.0):
    """batch['rays_i'] of the full-image branch of `build_rays` (lib/datasets/enerf_utils.py:25-31, 62-71), built on
    the device from the target camera: (B, h * w, 8) = [origin | direction | x, y] with h x w the size cv2.resize
    gives the scaled target image there (round(H*scale) x round(W*scale)); the size is computed HERE, once, and
    handed to the kernel."""
    B = tar_ext.shape[0]
    h, w = scaled_size(H, W, scale)
    rays = torch.empty(B, h * w, 8, device=tar_ext.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_make_rays(dptr(_c(tar_ext), "tar_ext"), dptr(_c(tar_ixt), "tar_ixt"), B, h, w,
                                 float(scale), dptr(rays), stream()), "make_rays")
    rays._bmv_built_rays = True      # built from the batch's camera, not handed over by the caller: rebuilt when reused
    return rays


def depth_values_uniform(near_far, D, h, w, depth_inv):
    B = near_far.shape[0]
    dv = torch.empty(B, D, h, w, device=near_far.device, dtype=torch.float32)
    nf = torch.empty(B, 2, h, w, device=near_far.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_depth_values_uniform(dptr(_c(near_far), "near_far"), B, D, h, w, int(bool(depth_inv)),
                                            dptr(dv), dptr(nf), stream()), "depth_values_uniform")
    return dv, nf


def frame_setup(src_exts, src_ixts, tar_ext, tar_ixt, src_scales, tar_scales, near_far, D, h, w, depth_inv):
    """proj_mats of every cascade level + depth_values_uniform of level 0 in ONE launch (what a frame needs from the
    cameras alone): ([proj_l (B,S,3,4)], (depth_values (B,D,h,w), near_far (B,2,h,w)))."""
    B, S = src_exts.shape[:2]
    L = len(src_scales)
    dev = src_exts.device
    proj = torch.empty(L, B, S, 3, 4, device=dev, dtype=torch.float32)
    dv = torch.empty(B, D, h, w, device=dev, dtype=torch.float32)
    nf = torch.empty(B, 2, h, w, device=dev, dtype=torch.float32)
    ss = (C.c_float * L)(*[float(v) for v in src_scales])
    ts = (C.c_float * L)(*[float(v) for v in tar_scales])
    lib = _lib.load()
    _lib.check(lib.bmv_frame_setup(dptr(_c(src_exts), "src_exts"), dptr(_c(src_ixts), "src_ixts"), dptr(_c(tar_ext), "tar_ext"),
                                   dptr(_c(tar_ixt), "tar_ixt"), ss, ts, L, B, S, dptr(proj), dptr(_c(near_far), "near_far"),
                                   D, h, w, int(bool(depth_inv)), dptr(dv), dptr(nf), stream()), "frame_setup")
    return [proj[l] for l in range(L)], (dv, nf)


def depth_values_cascade(depth, std, near_far, h, w, D):
    B, h0, w0 = depth.shape
    dv = torch.empty(B, D, h, w, device=depth.device, dtype=torch.float32)
    nf = torch.empty(B, 2, h, w, device=depth.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_depth_values_cascade(dptr(_c(depth), "depth"), dptr(_c(std), "std"),
                                            dptr(_c(near_far), "near_far"), B, h0, w0, h, w, D, dptr(dv), dptr(nf),
                                            stream()), "depth_values_cascade")
    return dv, nf


def homo_warp(src_feat, proj, depth_values, want_grid=True):
    B, C_, Hs, Ws = src_feat.shape
    _, D, h, w = depth_values.shape
    warped = torch.empty(B, C_, D, h, w, device=src_feat.device, dtype=torch.float32)
    grid = torch.empty(B, D, h, w, 2, device=src_feat.device, dtype=torch.float32) if want_grid else None
    lib = _lib.load()
    _lib.check(lib.bmv_homo_warp_fwd(dptr(_c(src_feat), "src_feat"), dptr(_c(proj), "proj"),
                                     dptr(_c(depth_values), "depth_values"), B, C_, Hs, Ws, D, h, w, dptr(warped),
                                     dptr(grid), stream()), "homo_warp")
    return warped, grid


def nchw_to_nhwc(x):
    """(..., C, H, W) -> (..., H, W, C) contiguous, C % 4 == 0 (HIP transpose kernel)."""
    C_, H, W = x.shape[-3:]
    n = x.numel() // (C_ * H * W)
    out = torch.empty(*x.shape[:-3], H, W, C_, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with ktimer.region(f"nchw_to_nhwc[C={C_},{H}x{W}]"):
        rc = lib.bmv_nchw_to_nhwc(dptr(_c(x), "x"), n, C_, H, W, dptr(out), stream())
    _lib.check(rc, "nchw_to_nhwc")
    return out


class QuadFeats:
    """Source feature maps in the plane sweep's QUAD-PLANAR layout (csrc/sweep_quad.hip): `data` is (N, C/4, H, W, 4) or
    (B, V, C/4, H, W, 4) contiguous -- a record of 4 channels = 16 bytes.  Written directly by the convolution engine in
    inference (convnet.conv_fwd(..., channels_last="quad")); `to_quad_planar` converts planar / channel-last maps.
    `shape` is the LOGICAL planar shape, `to_nchw()` the planar tensor (a copy)."""
    __slots__ = ("data",)

    def __init__(self, data):
        if data.dim() not in (5, 6) or data.shape[-1] != 4 or not data.is_contiguous():
            raise ValueError("QuadFeats wants a contiguous (..., C/4, H, W, 4) tensor")
        self.data = data

    @property
    def shape(self):            # the logical (N, C, H, W) / (B, V, C, H, W)
        *lead, Q, H, W, _ = self.data.shape
        return torch.Size((*lead, Q * 4, H, W))

    @property
    def device(self):
        return self.data.device

    @property
    def is_cuda(self):
        return self.data.is_cuda

    def dim(self):
        return self.data.dim() - 1

    def reshape_views(self, B, V):
        """(B V, C/4, H, W, 4) -> the (B, V, ...) form the sweep takes."""
        return QuadFeats(self.data.reshape(B, V, *self.data.shape[-4:]))

    def contiguous(self):       # (what `.contiguous()` on the earlier channel-last views gave: the planar tensor)
        return self.to_nchw()

    def to_nchw(self):
        *lead, Q, H, W, _ = self.data.shape
        n = len(lead)
        return self.data.permute(*range(n), n, n + 3, n + 1, n + 2).reshape(*lead, Q * 4, H, W)


def to_quad_planar(x, channels_last=False):
    """(..., C, H, W) planar -- or (..., H, W, C) with channels_last=True -- -> (..., C/4, H, W, 4), C % 4 == 0."""
    if channels_last:
        H, W, C_ = x.shape[-3:]
    else:
        C_, H, W = x.shape[-3:]
    n = x.numel() // (C_ * H * W)
    out = torch.empty(*x.shape[:-3], C_ // 4, H, W, 4, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with ktimer.region(f"to_quad_planar[C={C_},{H}x{W}]"):
        rc = lib.bmv_to_quad_planar(dptr(_c(x), "x"), int(bool(channels_last)), n, C_, H, W, dptr(out), stream())
    _lib.check(rc, "to_quad_planar")
    return out


def _plane_uniform(depth_values, plane_uniform):
    """dv_plane_uniform of include/bmv.h from the hypotheses' shape: (B,D) -> 1; (B,D,h,w) -> 0, or 2 when the caller
    vouches that the planes are constant (cascade level 0: `plane_uniform=True`)."""
    if depth_values.dim() == 2:
        return 1
    return 2 if plane_uniform else 0


class QuadVolume:
    """A cost volume as QUAD RECORDS: `data` is (B, C/4, D, h, w, 4) contiguous -- what the inference sweep writes with
    `quad_out=True` (one 16-byte store per voxel and channel quad, csrc/sweep_quad.hip) and the regulariser's first layer
    stages with one 16-byte load per position (convnet.conv_c4_fwd).  `to_planar()` gives (B, C, D, h, w)."""

    def __init__(self, data):
        if data.dim() != 6 or data.shape[-1] != 4 or not data.is_contiguous():
            raise ValueError("QuadVolume takes a contiguous (B, C/4, D, h, w, 4) tensor")
        self.data = data

    @property
    def shape(self):
        B, Q, D, h, w, _ = self.data.shape
        return torch.Size((B, 4 * Q, D, h, w))

    @property
    def is_cuda(self):
        return self.data.is_cuda

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def device(self):
        return self.data.device

    def to_planar(self):
        B, Q, D, h, w, _ = self.data.shape
        return self.data.permute(0, 1, 5, 2, 3, 4).reshape(B, 4 * Q, D, h, w)


QUAD_VOLUME_FLAG = 1 << 24      # include/bmv.h: bmv_sweep_variance_quad_fwd flags


def sweep_variance_quad(feats, proj, depth_values, view_ids=None, hw=None, plane_uniform=False, variant=-1, out=None,
                        flags=0, quad_out=False):
    """Plane sweep on quad-planar features (QuadFeats or its (B,V,C/4,Hs,Ws,4) tensor).  `view_ids` (B,S) int32 picks the
    S views of proj from the V views of feats; None: V == S.  flags: include/bmv.h (bits 16-23: LDS budget in KB)."""
    if sweep_hook is not None:
        r = sweep_hook(_sweep_variance_quad, (feats, proj, depth_values),
                       dict(view_ids=view_ids, hw=hw, plane_uniform=plane_uniform, variant=variant, out=out, flags=flags,
                            quad_out=quad_out))
        if r is not None:
            return r
    return _sweep_variance_quad(feats, proj, depth_values, view_ids, hw, plane_uniform, variant, out, flags, quad_out)


def _sweep_variance_quad(feats, proj, depth_values, view_ids=None, hw=None, plane_uniform=False, variant=-1, out=None,
                         flags=0, quad_out=False):
    """quad_out: the variance as a QuadVolume (quad records) when the kernel variant has that output, else planar."""
    data = feats.data if isinstance(feats, QuadFeats) else feats
    if data.dim() != 6:
        raise ValueError("the quad-planar sweep takes (B, V, C/4, Hs, Ws, 4) features (QuadFeats.reshape_views)")
    B, V, Q, Hs, Ws, _ = data.shape
    C_ = Q * 4
    S = proj.shape[1]
    if depth_values.dim() == 2:
        D = depth_values.shape[1]
        h, w = hw
    else:
        _, D, h, w = depth_values.shape
    if view_ids is not None:
        if view_ids.dtype != torch.int32 or tuple(view_ids.shape) != (B, S):
            raise ValueError("view_ids must be int32 (B,S)")
        check_view_ids(view_ids, V)
    elif V != S:
        raise ValueError(f"{V} source views but {S} projection matrices (pass view_ids to pick)")
    lib = _lib.load()
    if isinstance(out, QuadVolume) or (quad_out and out is None):
        given = isinstance(out, QuadVolume)        # (framegraph.py replays a sweep into the buffer of its first run)
        qv = out.data if given else torch.empty(B, Q, D, h, w, 4, device=data.device, dtype=torch.float32)
        with ktimer.region(f"sweep_variance[C={C_},D={D},{h}x{w}]", bind=True):
            rc = lib.bmv_sweep_variance_quad_fwd(dptr(data, "feats_quad"), dptr(_c(view_ids), "view_ids", torch.int32) if view_ids is not None else None,
                                                 V, dptr(_c(proj), "proj"), dptr(_c(depth_values), "depth_values"),
                                                 _plane_uniform(depth_values, plane_uniform), B, S, C_,
                                                 Hs, Ws, D, h, w, dptr(qv), int(variant), int(flags) | QUAD_VOLUME_FLAG, stream())
        if rc == 0:
            return out if given else QuadVolume(qv)
        if rc != BMV_ERR_UNSUPPORTED or given:
            _lib.check(rc, "sweep_variance_quad")
        lib.bmv_last_error()            # a variant without the quad-record output (tuning), or an uncovered shape: planar
        del qv
    if out is None:
        out = torch.empty(B, C_, D, h, w, device=data.device, dtype=torch.float32)
    with ktimer.region(f"sweep_variance[C={C_},D={D},{h}x{w}]", bind=True):
        rc = lib.bmv_sweep_variance_quad_fwd(dptr(data, "feats_quad"), dptr(_c(view_ids), "view_ids", torch.int32) if view_ids is not None else None,
                                             V, dptr(_c(proj), "proj"), dptr(_c(depth_values), "depth_values"),
                                             _plane_uniform(depth_values, plane_uniform), B, S, C_,
                                             Hs, Ws, D, h, w, dptr(out), int(variant), int(flags), stream())
    if rc == BMV_ERR_UNSUPPORTED and variant < 0 and isinstance(feats, QuadFeats):
        # a shape outside the planned kernel's 32-bit / 24-bit offset budget (e.g. 1440 x 2560 inputs: level 0 is
        # 360 x 640 x 64), or more than 64 channels: the reference-layout direct-gather kernel (algo 1), views gathered
        # by index -- slow and correct instead of a failed forward
        _lib.load().bmv_last_error()
        nchw = feats.to_nchw()                                        # (B, V, C, Hs, Ws)
        if view_ids is not None:
            nchw = torch.gather(nchw, 1, view_ids.long()[:, :, None, None, None].expand(-1, -1, *nchw.shape[2:]))
        dv = depth_values if depth_values.dim() == 4 else depth_values[:, :, None, None].expand(B, D, h, w).contiguous()
        return _sweep_variance(nchw.contiguous(), proj, dv, 1, out, False, False)
    _lib.check(rc, "sweep_variance_quad")
    return out


sweep_hook = None     # framegraph.FrameGraph: hook(impl, args, kwargs) lets a graph capture step around one sweep launch


def sweep_variance(feats, proj, depth_values, algo=0, out=None, channels_last=None, plane_uniform=False, quad_out=False):
    """feats: QuadFeats (the inference layout: csrc/sweep_quad.hip), or (B,S,C,Hs,Ws) in the reference layout, or
    channel-last (B,S,Hs,Ws,C) when channels_last=True.  With channels_last=None (default) a reference-layout input
    with C in {16, 32} is first put into the channel-last layout (one transpose kernel) and the windowed channel-last
    sweep runs; algo=1 forces the reference-layout direct-gather kernel; algo 500 + i / 600 + i: the quad-planar kernel
    (tuning variant i; 600: hypotheses declared plane-uniform) after a layout conversion.  `plane_uniform`: the caller
    vouches that every plane of depth_values is constant (cascade level 0).  `quad_out`: the variance as a QuadVolume
    (quad records, for convnet.conv_c4_fwd) where the quad-planar kernel runs; planar otherwise."""
    if sweep_hook is not None:
        r = sweep_hook(_sweep_variance, (feats, proj, depth_values),
                       dict(algo=algo, out=out, channels_last=channels_last, plane_uniform=plane_uniform, quad_out=quad_out))
        if r is not None:
            return r
    return _sweep_variance(feats, proj, depth_values, algo, out, channels_last, plane_uniform, quad_out)


def mark_view_ids(view_ids, n_all):
    """The caller vouches that every entry of `view_ids` is in [0, n_all) (the K-volume networks: the indices are rows
    of combinations(range(N), 3) picked by host-validated triplet numbers) -- no device read."""
    view_ids._bmv_view_range = int(n_all)
    return view_ids


def check_view_ids(view_ids, n_all):
    """The kernels index the all-views buffers with these ids: make sure they are in [0, n_all).  Costs one device ->
    host read of (min, max) unless the tensor is already marked (mark_view_ids / an earlier check); cannot be done
    inside a stream capture -- validate before capturing."""
    if getattr(view_ids, "_bmv_view_range", None) == int(n_all):
        return view_ids
    if view_ids.is_cuda and torch.cuda.is_current_stream_capturing():
        raise ValueError("view_ids reach a kernel unvalidated inside a stream capture: call ops.check_view_ids(view_ids, "
                         "n_all) before capturing")
    if view_ids.numel():
        lo, hi = (int(v) for v in view_ids.aminmax())
        if lo < 0 or hi >= int(n_all):
            raise ValueError(f"view_ids must lie in [0, {int(n_all)}): found [{lo}, {hi}] (a bad view_selection.json?)")
    return mark_view_ids(view_ids, n_all)


def sweep_variance_views(feats_all, view_ids, proj, depth_values, out=None, plane_uniform=False, quad_out=False):
    """Plane sweep over the S views `view_ids` (B,S) int32 picked from feats_all = ALL source views: QuadFeats (what
    FeatureNet's engine path returns in inference), or a (B,n_all,C,Hs,Ws) view of a channel-last (B,n_all,Hs,Ws,C)
    buffer.  No gathered copy of the feature maps is made."""
    if sweep_hook is not None:
        r = sweep_hook(_sweep_variance_views, (feats_all, view_ids, proj, depth_values),
                       dict(out=out, plane_uniform=plane_uniform, quad_out=quad_out))
        if r is not None:
            return r
    return _sweep_variance_views(feats_all, view_ids, proj, depth_values, out, plane_uniform, quad_out)


def _sweep_variance_views(feats_all, view_ids, proj, depth_values, out=None, plane_uniform=False, quad_out=False):
    if isinstance(feats_all, QuadFeats):
        return _sweep_variance_quad(feats_all, proj, depth_values, view_ids, None, plane_uniform, -1, out, 0, quad_out)
    cl = feats_all.permute(0, 1, 3, 4, 2) if feats_all.dim() == 5 else None
    if cl is None or not cl.is_contiguous():
        raise ValueError("sweep_variance_views needs a (B,n_all,C,Hs,Ws) view of channel-last feature maps")
    B, n_all, Hs, Ws, C_ = cl.shape
    S = view_ids.shape[1]
    if view_ids.dtype != torch.int32 or view_ids.shape[0] != B:
        raise ValueError("view_ids must be int32 (B,S)")
    check_view_ids(view_ids, n_all)
    _, D, h, w = depth_values.shape
    if out is None:
        out = torch.empty(B, C_, D, h, w, device=cl.device, dtype=torch.float32)
    lib = _lib.load()
    with ktimer.region(f"sweep_variance[C={C_},D={D},{h}x{w}]", bind=True):
        rc = lib.bmv_sweep_variance_views_fwd(dptr(cl, "feats_all"), dptr(_c(view_ids), "view_ids", torch.int32), n_all,
                                              dptr(_c(proj), "proj"), dptr(_c(depth_values), "depth_values"), B, S, C_,
                                              Hs, Ws, D, h, w, dptr(out), stream())
    _lib.check(rc, "sweep_variance_views")
    return out


def _sweep_variance(feats, proj, depth_values, algo=0, out=None, channels_last=None, plane_uniform=False, quad_out=False):
    if isinstance(feats, QuadFeats) and not (algo == 0 or algo >= 500):
        feats, channels_last = feats.to_nchw(), None     # another kernel was asked for by id: back to the planar layout
    if isinstance(feats, QuadFeats) or algo >= 500:      # quad-planar kernel (500 + i: tuning variant i; 600 + i: the same
        variant = -1                                     # with the hypotheses declared plane-uniform)
        pu = bool(plane_uniform)
        if algo >= 600:
            variant, pu = algo - 600, True
        elif algo >= 500:
            variant = algo - 500
        if not isinstance(feats, QuadFeats):
            if channels_last is None and feats.dim() == 5 and not feats.is_contiguous():
                cl = feats.permute(0, 1, 3, 4, 2)
                if cl.is_contiguous():
                    feats, channels_last = cl, True
            feats = QuadFeats(to_quad_planar(feats, channels_last=bool(channels_last)))
        return _sweep_variance_quad(feats, proj, depth_values, None, None, pu, variant, out, 0, quad_out)
    if channels_last is None and feats.dim() == 5 and not feats.is_contiguous() and algo != 1:
        cl = feats.permute(0, 1, 3, 4, 2)
        if cl.is_contiguous():        # (B,S,C,Hs,Ws) view of a channel-last buffer (the conv engine writes it so)
            feats, channels_last = cl, True
    if channels_last:
        B, S, Hs, Ws, C_ = feats.shape
    else:
        B, S, C_, Hs, Ws = feats.shape
    _, D, h, w = depth_values.shape
    if out is None:
        out = torch.empty(B, C_, D, h, w, device=feats.device, dtype=torch.float32)
    layout = 1 if channels_last else 0
    if channels_last is None and algo != 1 and C_ in (16, 32):
        feats = nchw_to_nhwc(feats)
        layout = 1
    lib = _lib.load()
    args = (dptr(_c(feats), "feats"), dptr(_c(proj), "proj"), dptr(_c(depth_values), "depth_values"), B, S, C_, Hs,
            Ws, D, h, w, dptr(out), layout, int(algo), stream())
    with ktimer.region(f"sweep_variance[C={C_},D={D},{h}x{w}]", bind=True):
        rc = lib.bmv_sweep_variance_fwd(*args)
    _lib.check(rc, "sweep_variance")
    return out


def depth_regress(depth_prob, depth_values, depth_inv, frame_outputs=()):
    """frame_outputs: which of ("depth", "std") are outputs of the frame being captured as they are (autograph): the
    kernel then ALSO writes them to the caller's tensors through the pointer table (no copy node at the frame's end)."""
    B, D, h, w = depth_values.shape
    depth = torch.empty(B, h, w, device=depth_values.device, dtype=torch.float32)
    std = torch.empty_like(depth)
    lib = _lib.load()
    if frame_outputs and _deferring():
        if "depth" in frame_outputs:
            defer_output(depth, fresh=True)
        if "std" in frame_outputs:
            defer_output(std, fresh=True)
    _lib.check(lib.bmv_depth_regress_fwd(dptr(_c(depth_prob), "depth_prob"), dptr(_c(depth_values), "depth_values"),
                                         B, D, h, w, int(bool(depth_inv)), dptr(depth), dptr(std), stream()),
               "depth_regress")
    return depth, std


def build_rays(rays, depth, std, near_far, Hr, Wr, depth_inv):
    B, N = rays.shape[:2]
    hv, wv = depth.shape[-2:]
    out = torch.empty(B, N, 12, device=rays.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_build_rays(dptr(_c(rays), "rays"), dptr(_c(depth), "depth"), dptr(_c(std), "std"),
                                  dptr(_c(near_far), "near_far"), B, N, hv, wv, Hr, Wr, int(bool(depth_inv)),
                                  dptr(out), stream()), "build_rays")
    return out


def sample_along_depth(rays, Ns, depth_inv):
    B, N = rays.shape[:2]
    xyz = torch.empty(B, N, Ns, 3, device=rays.device, dtype=torch.float32)
    uvd = torch.empty(B, N, Ns, 3, device=rays.device, dtype=torch.float32)
    z = torch.empty(B, N, Ns, device=rays.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_sample_along_depth(dptr(_c(rays), "rays"), B, N, Ns, int(bool(depth_inv)), dptr(xyz),
                                          dptr(uvd), dptr(z), stream()), "sample_along_depth")
    return xyz, uvd, z


def unpreprocess(src, Ho, Wo):
    """src (B,S,3,H,W) in [-1,1] -> (B,S,3,Ho,Wo)."""
    B, S, C_, H, W = src.shape
    out = torch.empty(B, S, C_, Ho, Wo, device=src.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_unpreprocess(dptr(_c(src), "src"), B * S, C_, H, W, Ho, Wo, dptr(out), stream()),
               "unpreprocess")
    return out


def vox_feat(uvd01, volume):
    B, P = uvd01.shape[:2]
    _, C_, D, h, w = volume.shape
    out = torch.empty(B, P, C_, device=volume.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_vox_feat(dptr(_c(uvd01), "uvd"), dptr(_c(volume), "volume"), B, P, C_, D, h, w, dptr(out),
                                stream()), "vox_feat")
    return out


def img_feat(xyz, img_feat_rgb, src_exts, src_ixts, tar_ext, render_scale):
    B, S, C_, H, W = img_feat_rgb.shape
    pts = _c(xyz).reshape(B, -1, 3)
    P = pts.shape[1]
    out = torch.empty(B, P, S, C_ + 4, device=xyz.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_img_feat(dptr(pts, "xyz"), dptr(_c(img_feat_rgb), "img_feat_rgb"),
                                dptr(_c(src_exts), "src_exts"), dptr(_c(src_ixts), "src_ixts"),
                                dptr(_c(tar_ext), "tar_ext"), float(render_scale), B, P, S, C_, H, W, dptr(out),
                                stream()), "img_feat")
    return out


NERF_PARAM_ORDER = ("agg.view_fc.0", "agg.global_fc.0", "agg.agg_w_fc.0", "agg.fc.0", "lr0.0", "sigma.0",
                    "color.0", "color.2")


def nerf_pack_weights(tensors, feat_ch, out=None):
    """tensors: 16 CUDA tensors (weight, bias of the 8 Linear layers in NERF_PARAM_ORDER)."""
    lib = _lib.load()
    n = lib.bmv_nerf_blob_size(int(feat_ch))
    if n < 0:
        _lib.check(n, "nerf_blob_size")
    if out is None:
        out = torch.empty(n, device=tensors[0].device, dtype=torch.float32)
    held = [_c(t.detach()) for t in tensors]
    params = _lib.NerfParams(*[dptr(t, f"nerf param {i}") for i, t in enumerate(held)])
    _lib.check(lib.bmv_nerf_pack_weights(C.byref(params), int(feat_ch), dptr(out), stream()), "nerf_pack_weights")
    return out


def nerf_mlp(vox_feat_t, img_feat_rgb_dir, blob, feat_ch):
    """vox_feat (..., 8), img_feat_rgb_dir (..., S, feat_ch + 7) with S in {2, 3, 4} source views -> (..., 4)."""
    lead = vox_feat_t.shape[:-1]
    npts = vox_feat_t.numel() // 8
    S = int(img_feat_rgb_dir.shape[-2])
    out = torch.empty(*lead, 4, device=vox_feat_t.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_nerf_mlp_fwd(dptr(_c(vox_feat_t), "vox_feat"), dptr(_c(img_feat_rgb_dir), "img_feat_rgb_dir"),
                                    dptr(blob, "blob"), int(feat_ch), S, npts, dptr(out), stream()), "nerf_mlp")
    return out


def composite(raw, z_vals, white_bkgd=False):
    lead = raw.shape[:-2]
    Ns = raw.shape[-2]
    nrays = raw.numel() // (Ns * 4)
    rgb = torch.empty(*lead, 3, device=raw.device, dtype=torch.float32)
    depth = torch.empty(*lead, device=raw.device, dtype=torch.float32)
    weights = torch.empty(*lead, Ns, device=raw.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_composite_fwd(dptr(_c(raw), "raw"), dptr(_c(z_vals), "z_vals"), nrays, Ns,
                                     int(bool(white_bkgd)), dptr(rgb), dptr(depth), dptr(weights), stream()),
               "composite")
    return rgb, depth, weights


def mask_viewport(xyz, src_exts, src_ixts, inv_w, inv_h):
    B = xyz.shape[0]
    pts = _c(xyz).reshape(B, -1, 3)
    P = pts.shape[1]
    V = src_exts.shape[1]
    mask = torch.empty(B, P, device=xyz.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_mask_viewport(dptr(pts, "xyz"), dptr(_c(src_exts), "src_exts"), dptr(_c(src_ixts), "src_ixts"),
                                     float(inv_w), float(inv_h), B, P, V, dptr(mask), stream()), "mask_viewport")
    return mask


def ndc_coords(xyz, src_ext, src_ixt, inv_w, inv_h):
    """get_ndc_coords (enerf/utils.py:490-508): xyz (B,N,Ns,3), one source view per item -> (B,N,Ns,3)."""
    B = xyz.shape[0]
    pts = _c(xyz).reshape(B, -1, 3)
    out = torch.empty_like(pts)
    lib = _lib.load()
    _lib.check(lib.bmv_ndc_coords(dptr(pts, "xyz"), dptr(_c(src_ext).reshape(B, 16), "src_ext"),
                                  dptr(_c(src_ixt).reshape(B, 9), "src_ixt"), float(inv_w), float(inv_h), B,
                                  pts.shape[1], dptr(out), stream()), "ndc_coords")
    return out.reshape(xyz.shape)


def blend(raws, masks, z_vals, normalise):
    B, K, N, Ns = raws.shape[:4]
    rgb = torch.empty(B, N, 3, device=raws.device, dtype=torch.float32)
    depth = torch.empty(B, N, device=raws.device, dtype=torch.float32)
    weights = torch.empty(B, N, Ns, device=raws.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_blend_fwd(dptr(_c(raws), "raws"), dptr(_c(masks).reshape(B, K, N, Ns), "masks"),
                                 dptr(_c(z_vals), "z_vals"), B, K, N, Ns, int(bool(normalise)), dptr(rgb),
                                 dptr(depth), dptr(weights), stream()), "blend")
    return rgb, depth, weights


def render_rays(rays, depth, std, near_far, volume, im_feat, rgb_src, src_exts, src_ixts, tar_ext, blob, *, feat_ch,
                Ns, depth_inv, Hr, Wr, render_scale, rgb_affine, white_bkgd=False, mode=0, ray_range=None, outs=None,
                view_ids=None, im_packed=None):
    """Fused a6..a12 (+a14).  mode 0 -> (rgb, depth, weights); mode 1 -> (raw, z_vals, mask).
    ray_range=(begin, end) renders only those rays; the outputs keep the full (B,N,...) shape and
    only that slice is written.  view_ids (B,S) int32: im_feat / rgb_src hold ALL n_all source views and the
    cost volume's view i is view_ids[b, i] (src_exts / src_ixts are already the S picked ones).
    im_packed (B, S | n_all, Hr, Wr, 12): the views' lookup records (convnet.LookupRecords) instead of im_feat /
    rgb_src (which may then be None): one 16-byte + one 8-byte load per tap and lane half."""
    B, N = rays.shape[:2]
    S = src_exts.shape[1]
    vol_packed = not torch.is_tensor(volume)   # convnet.VolumeRecords: (B,Dv,hv,wv,8) voxel records
    if vol_packed:
        if im_packed is None:
            raise ValueError("render_rays: volume records come with image records (im_packed)")
        volume = volume.t
        _, Dv, hv, wv, _ = volume.shape
    else:
        _, _, Dv, hv, wv = volume.shape
    dev = rays.device
    if outs is not None:
        o0, o1, o2 = outs
        want = ((B, N, 3), (B, N), (B, N, Ns)) if mode == 0 else ((B, N, Ns, 4), (B, N, Ns), (B, N, Ns))
        for t, shp in zip(outs, want):
            if tuple(t.shape) != shp:
                raise ValueError(f"render_rays: output buffer {tuple(t.shape)} != {shp}")
    elif mode == 0:
        o0 = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
        o1 = torch.empty(B, N, device=dev, dtype=torch.float32)
        o2 = torch.empty(B, N, Ns, device=dev, dtype=torch.float32)
    else:
        o0 = torch.empty(B, N, Ns, 4, device=dev, dtype=torch.float32)
        o1 = torch.empty(B, N, Ns, device=dev, dtype=torch.float32)
        o2 = torch.empty(B, N, Ns, device=dev, dtype=torch.float32)
    if im_packed is not None:
        if tuple(im_packed.shape[-3:]) != (Hr, Wr, 12) or int(feat_ch) != 8 or im_packed.dim() != 5:
            raise ValueError(f"im_packed {tuple(im_packed.shape)}: expected (B,views,{Hr},{Wr},12) records and feat_ch 8")
        n_views = im_packed.shape[1]
        names = ("rays", "depth", "std", "near_far", "volume", "src_exts", "src_ixts", "tar_ext", "im_packed")
        held = [_c(t) for t in (rays, depth, std, near_far, volume, src_exts, src_ixts, tar_ext, im_packed)]
    else:
        if tuple(im_feat.shape[-2:]) != (Hr, Wr) or tuple(rgb_src.shape[-2:]) != (Hr, Wr):
            raise ValueError(f"im_feat {tuple(im_feat.shape)} / rgb_src {tuple(rgb_src.shape)} must be at the render "
                             f"resolution ({Hr},{Wr})")
        n_views = im_feat.shape[1]
        names = ("rays", "depth", "std", "near_far", "volume", "im_feat", "rgb_src", "src_exts", "src_ixts", "tar_ext")
        held = [_c(t) for t in (rays, depth, std, near_far, volume, im_feat, rgb_src, src_exts, src_ixts, tar_ext)]
    begin, end = ray_range if ray_range is not None else (0, N)
    a = _lib.RenderArgs()
    for name, t in zip(names, held):
        setattr(a, name, dptr(t, name))
    a.blob = dptr(blob, "blob")
    a.B, a.N, a.S, a.feat_ch, a.Ns, a.depth_inv = B, N, S, int(feat_ch), int(Ns), int(bool(depth_inv))
    a.hv, a.wv, a.Dv, a.Hr, a.Wr = hv, wv, Dv, int(Hr), int(Wr)
    a.render_scale = float(render_scale)
    a.rgb_affine, a.white_bkgd, a.mode = int(bool(rgb_affine)), int(bool(white_bkgd)), int(mode)
    a.ray_begin, a.ray_end = int(begin), int(end)
    a.out0, a.out1, a.out2 = dptr(o0), dptr(o1), dptr(o2)
    a.vol_packed = int(vol_packed)
    if view_ids is not None:
        if view_ids.dtype != torch.int32 or tuple(view_ids.shape) != (B, S) or (
                im_packed is None and im_feat.shape[1] != rgb_src.shape[1]):
            raise ValueError("render_rays: view_ids must be int32 (B,S); im_feat / rgb_src must hold the same n_all views")
        check_view_ids(view_ids, n_views)
        view_ids = _c(view_ids)
        a.view_ids, a.n_all = dptr(view_ids, "view_ids", torch.int32), int(n_views)
    else:
        a.view_ids, a.n_all = None, 0
    lib = _lib.load()
    if defer_table is not None:
        defer_input(held[0])
        if mode == 0:
            for o in (o0, o1, o2):
                defer_output(o, fresh=outs is None)
    with ktimer.region(f"render_rays[feat={feat_ch},Ns={Ns},mode={mode}]"):
        rc = lib.bmv_render_rays_fwd(C.byref(a), stream())
    _lib.check(rc, "render_rays")
    return o0, o1, o2


# ======================================================================= deferred pointers (autograph)
class PtrTable:
    """A device table of pointers that a CAPTURED frame reads its large inputs and writes its large outputs through
    (include/bmv.h, bmv_defer_pointer): the graph bakes the table's address in, `set()` points the entries at this
    frame's tensors before a replay -- no copy into captured input buffers, none out of captured output buffers.
    Inputs are registered by autograph (`add_input(static clone)`); outputs register themselves while the frame is
    captured (`render_rays` for buffers it allocates).  Only while `ops.defer_table` is set AND the current stream is
    capturing do the wrappers defer anything."""
    SLOTS = 16

    def __init__(self, device):
        self.t = torch.zeros(self.SLOTS, dtype=torch.int64, device=device)
        self.inputs = {}          # data_ptr of the static tensor -> slot
        self.outputs = {}         # data_ptr of the static output -> (slot, tensor)
        self.n = 0
        self.taken = set()        # slots a launch actually deferred (an input nobody deferred keeps its copy)
        self.ring = None          # FeedRing, when the frame's first node reads its arguments from the host ring

    def _slot(self):
        if self.n >= self.SLOTS:
            raise RuntimeError("PtrTable: out of slots")
        self.n += 1
        return self.n - 1

    def add_input(self, t):
        slot = self._slot()
        self.inputs[t.data_ptr()] = slot
        return slot

    def set(self, slots, ptrs):
        n = len(slots)
        if n == 0:
            return
        _lib.check(_lib.load().bmv_ptr_table_set(self.t.data_ptr(), n, (C.c_int * n)(*slots), (C.c_void_p * n)(*ptrs),
                                                 stream()), "ptr_table_set")


class FeedRing:
    """The arguments of a captured frame's first node (bmv_frame_feed_ring): R messages in pinned host memory.  The host
    writes message number `posted` (plain stores through a numpy view) and replays the graph; the node's n-th
    execution reads message n.  EVERY replay of a graph that contains the node must be preceded by exactly one post."""
    R = 64
    DT = [("seq", "<u4"), ("n_ptr", "<i4"), ("n_copy", "<i4"), ("pad", "<i4"), ("slot", "<i4", 16), ("value", "<u8", 16),
          ("src", "<u8", 8), ("dst", "<u8", 8), ("count", "<i4", 8)]

    def __init__(self, device):
        import numpy as np
        dt = np.dtype(self.DT)
        assert dt.itemsize == _lib.load().bmv_frame_feed_msg_bytes()
        self.host = torch.zeros(self.R * dt.itemsize, dtype=torch.uint8).pin_memory()
        self.msgs = self.host.numpy().view(dt)
        self.state = torch.zeros(2, dtype=torch.int32, device=device)     # [executions of the node, sequence faults]
        self.posted = 0
        self.pending = False      # a message has been posted and its replay not yet reported
        self.fast = None
        self.events = {}          # block number -> event recorded behind the last replay of that block of R / 2 messages

    # A caller that does not synchronise per frame can run ahead of the GPU by more frames than the ring has slots (a
    # graph launch only queues): slot n % R must not be rewritten before replay n - R has read it.  Every R / 2
    # replays an event is recorded; the first post of a block waits for the event of the block before the last.
    def _reserve(self):
        if self.pending:          # the last message was never replayed (an exception, a caller that changed its mind):
            self.posted -= 1      # it is superseded, not skipped -- execution n of the node reads message n
        self.pending = True
        n, half = self.posted, self.R // 2
        if n % half == 0 and n >= self.R:
            ev = self.events.pop(n // half - 2, None)
            if ev is not None:
                ev.synchronize()
            else:                 # (a replay nobody reported: be safe)
                torch.cuda.current_stream().synchronize()
            self.raise_on_faults()        # (already waiting on the device here: the read is nearly free)

    def replayed(self):
        """Call after the replay that followed a post."""
        self.pending = False
        n, half = self.posted - 1, self.R // 2
        if n % half == half - 1:
            ev = torch.cuda.Event()
            ev.record()
            self.events[n // half] = ev

    def node(self, table):
        """The node itself, on the current stream (autograph issues it first thing in the captured frame)."""
        _lib.check(_lib.load().bmv_frame_feed_ring(table.t.data_ptr(), self.host.data_ptr(), self.state.data_ptr(), self.R,
                                                   stream()), "frame_feed_ring")

    def post(self, slots=(), values=(), srcs=(), dsts=(), counts=()):
        self._reserve()
        m = self.msgs[self.posted % self.R]
        n, c = len(slots), len(srcs)
        m["seq"], m["n_ptr"], m["n_copy"] = self.posted & 0xffffffff, n, c
        if n:
            m["slot"][:n] = slots
            m["value"][:n] = values
        if c:
            m["src"][:c] = srcs
            m["dst"][:c] = dsts
            m["count"][:c] = counts
        self.posted += 1
        self.fast = None          # (the constant fields of the fast path are no longer in every slot)

    def prepare_fast(self, slots, dsts, counts):
        """The steady state posts the same slots / destinations / counts every time: written into all R messages
        once, a post then stores the sequence number, the table values and the copy sources only."""
        n, c = len(slots), len(dsts)
        # Messages posted and not yet consumed are about to be rewritten (a caller that does not synchronise per frame
        # has replays queued; after a slow post() one of them carries n_copy = 0 and must keep it: ADVICE r4) -- wait for
        # the device first.  Only at a slow -> fast transition, never in the steady state.
        if self.posted:
            torch.cuda.synchronize()
        for m in self.msgs:
            m["n_ptr"], m["n_copy"] = n, c
            if n:
                m["slot"][:n] = slots
            if c:
                m["dst"][:c] = dsts
                m["count"][:c] = counts
        self.fast = (n, c)

    def post_fast(self, values, srcs):
        self._reserve()
        m = self.msgs[self.posted % self.R]
        n, c = self.fast
        m["seq"] = self.posted & 0xffffffff
        if n:
            m["value"][:n] = values
        if c:
            m["src"][:c] = srcs
        self.posted += 1

    def faults(self):
        return int(self.state[1].item())

    def raise_on_faults(self):
        """state[1] counts frames whose first node found a message with another sequence number than its own execution
        count (it ran on a stale pointer table / stale small inputs): a silently wrong frame otherwise."""
        n = self.faults()
        if n:
            raise RuntimeError(f"FeedRing: {n} captured frame(s) ran on a message that was not theirs (sequence mismatch): "
                               "the frames since the last check are wrong")


def copy_to_slot(t, table, slot):
    """`t` (contiguous fp32) -> the tensor `table[slot]` points at when the kernel runs (bmv_copy_to_slot)."""
    _lib.check(_lib.load().bmv_copy_to_slot(dptr(t, "copy_to_slot"), table.t.data_ptr(), int(slot), t.numel(), stream()),
               "copy_to_slot")


def copy_to_slots(tensors, table, slots):
    """... up to 8 of them in one launch (bmv_copy_to_slots)."""
    n = len(tensors)
    _lib.check(_lib.load().bmv_copy_to_slots(n, (C.c_void_p * n)(*[dptr(t, "copy_to_slots").value for t in tensors]),
                                             table.t.data_ptr(), (C.c_int * n)(*[int(s) for s in slots]),
                                             (C.c_long * n)(*[t.numel() for t in tensors]), stream()), "copy_to_slots")


def defer_small_outputs(tensors):
    """Outputs of the frame being captured that no kernel writes through the table (small maps): copied to their
    table entries by ONE node of the frame's graph; autograph calls this at the end of the captured frame.  (Tried: the
    network placing it on its side stream under the renderer -- the fork / join around the persistent renderer cost
    more than the 4.5 us of the copy: fresh-tensor frame +8 us.)"""
    tb = defer_table
    if tb is None or not torch.cuda.is_current_stream_capturing():
        return
    todo = []
    for t in tensors:
        if (t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() not in tb.outputs and tb.n < tb.SLOTS
                and len(todo) < 8):
            slot = tb._slot()
            tb.outputs[t.data_ptr()] = (slot, t)
            todo.append((t, slot))
    if todo:
        copy_to_slots([t for t, _ in todo], tb, [s for _, s in todo])


defer_table = None      # the PtrTable of the capture in progress (autograph sets it around FrameGraph's capture)


def _deferring():
    return defer_table is not None and torch.cuda.is_current_stream_capturing()


def defer_input(t):
    """Call right before the launch that reads `t` (contiguous): if `t` is a registered static input of the frame being
    captured, that launch reads it through the table."""
    if defer_table is None or not torch.cuda.is_current_stream_capturing():
        return
    slot = defer_table.inputs.get(t.data_ptr())
    if slot is not None:
        rc = _lib.load().bmv_defer_pointer(t.data_ptr(), defer_table.t.data_ptr(), slot)
        if rc:                      # (not through check() on success: it would let go of the launch's held tensors)
            _lib.check(rc, "defer_pointer")
        defer_table.taken.add(slot)


def defer_output(t, fresh):
    """... the launch that WRITES `t`: buffers a wrapper allocated itself (`fresh`) are registered on their first
    launch, later launches into the same buffer (ray chunks) defer it again."""
    if defer_table is None or not torch.cuda.is_current_stream_capturing():
        return
    ent = defer_table.outputs.get(t.data_ptr())
    if ent is None:
        if not fresh:
            return
        ent = defer_table.outputs[t.data_ptr()] = (defer_table._slot(), t)
    rc = _lib.load().bmv_defer_pointer(t.data_ptr(), defer_table.t.data_ptr(), ent[0])
    if rc:
        _lib.check(rc, "defer_pointer")


# ======================================================================= MVSNeRF backbone
def mvs_proj_mats(src_exts, src_ixts):
    B, S = src_exts.shape[:2]
    out = torch.empty(B, S, 3, 4, device=src_exts.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_mvs_proj_mats(dptr(_c(src_exts), "src_exts"), dptr(_c(src_ixts), "src_ixts"), B, S, dptr(out),
                                     stream()), "mvs_proj_mats")
    return out


def resize_bilinear(x, h, w):
    """F.interpolate(x, (h, w), mode='bilinear', align_corners=False) on (..., C, H, W)."""
    C_, H, W = x.shape[-3:]
    n = x.numel() // (C_ * H * W)
    out = torch.empty(*x.shape[:-2], h, w, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_resize_bilinear(dptr(_c(x), "x"), n, C_, H, W, h, w, dptr(out), stream()), "resize_bilinear")
    return out


def mvs_sweep(imgs_small, feats, proj, depth_values, pad, algo=0):
    """a19 + a20.  feats (B,S,C,h,w): a reference-layout tensor, or a (B,S,C,h,w) VIEW of a channel-last (B,S,h,w,C)
    buffer (what MVSNeRF's FeatureNet returns on the engine path).  algo 0: the channel-last kernel (a planar input is
    converted first: one transpose launch); algo 1: the reference-layout gather kernel of round 1."""
    B, S, C_, h, w = feats.shape
    D = depth_values.shape[1]
    out = torch.empty(B, 3 * S + C_, D, h + 2 * pad, w + 2 * pad, device=feats.device, dtype=torch.float32)
    lib = _lib.load()
    cl = feats.permute(0, 1, 3, 4, 2)
    if algo == 0 and C_ == 32 and S == 3:
        if not cl.is_contiguous():
            cl = nchw_to_nhwc(feats)
        args = (dptr(_c(imgs_small), "imgs"), dptr(cl, "feats_cl"), dptr(_c(proj), "proj"),
                dptr(_c(depth_values), "depth_values"), B, S, C_, h, w, D, int(pad), dptr(out), stream())
        with ktimer.region(f"mvs_sweep[C={C_},D={D},{h}x{w},pad={pad}]"):
            rc = lib.bmv_mvs_sweep_cl_fwd(*args)
        _lib.check(rc, "mvs_sweep_cl")
        return out
    args = (dptr(_c(imgs_small), "imgs"), dptr(_c(feats), "feats"), dptr(_c(proj), "proj"),
            dptr(_c(depth_values), "depth_values"), B, S, C_, h, w, D, int(pad), dptr(out), stream())
    with ktimer.region(f"mvs_sweep[C={C_},D={D},{h}x{w},pad={pad}]"):
        rc = lib.bmv_mvs_sweep_fwd(*args)
    _lib.check(rc, "mvs_sweep")
    return out


MVS_MLP_PARAM_ORDER = tuple([f"pts_linears.{i}" for i in range(6)]) + ("pts_bias", "views_linears.0", "feature_linear",
                                                                        "alpha_linear", "rgb_linear")


def mvs_mlp_pack_weights(weights, biases):
    """weights / biases: dicts name -> tensor for the names in MVS_MLP_PARAM_ORDER."""
    lib = _lib.load()
    held = {k: (_c(weights[k].detach()), _c(biases[k].detach())) for k in MVS_MLP_PARAM_ORDER}
    p = _lib.MvsMlpParams()
    for i in range(6):
        p.pts_w[i] = held[f"pts_linears.{i}"][0].data_ptr()
        p.pts_b[i] = held[f"pts_linears.{i}"][1].data_ptr()
    for field, name in (("bias", "pts_bias"), ("views", "views_linears.0"), ("feature", "feature_linear"),
                        ("alpha", "alpha_linear"), ("rgb", "rgb_linear")):
        setattr(p, field + "_w", dptr(held[name][0], name + ".weight"))
        setattr(p, field + "_b", dptr(held[name][1], name + ".bias"))
    for i in range(6):
        dptr(held[f"pts_linears.{i}"][0], f"pts_linears.{i}.weight")   # validation only
    dev = held["pts_bias"][0].device
    blob = torch.empty(lib.bmv_mvs_mlp_blob_size(), device=dev, dtype=torch.float32)
    _lib.check(lib.bmv_mvs_mlp_pack_weights(C.byref(p), dptr(blob), stream()), "mvs_mlp_pack_weights")
    return blob


def _mvs_mlp_params(tensors):
    """22 tensors in the member order of bmv_mvs_mlp_params (6 pts weights, 6 pts biases, then weight / bias of
    pts_bias, views_linears.0, feature_linear, alpha_linear, rgb_linear) -> the C struct."""
    p = _lib.MvsMlpParams()
    for i in range(6):
        p.pts_w[i] = dptr(tensors[i], f"pts_linears.{i}.weight")
        p.pts_b[i] = dptr(tensors[6 + i], f"pts_linears.{i}.bias")
    for j, field in enumerate(("bias_w", "bias_b", "views_w", "views_b", "feature_w", "feature_b", "alpha_w", "alpha_b",
                               "rgb_w", "rgb_b")):
        setattr(p, field, dptr(tensors[12 + j], field))
    return p


MVS_MLP_SHAPES = ([(128, 63)] + [(128, 128)] * 4 + [(128, 191)] + [(128,)] * 6
                  + [(128, 20), (128,), (64, 131), (64,), (128, 128), (128,), (1, 128), (1,), (3, 64), (3,)])


def mvs_mlp_train_fwd(x, params):
    """Renderer_ours.forward for autograd (csrc/mvs_mlp_train.hip): x (npts, 86), params = the 22 parameter tensors in
    bmv_mvs_mlp_params order.  Returns (out (npts,4), act, scratch): `act` holds every layer's input and
    pre-activation for ONE mvs_mlp_train_bwd."""
    lib = _lib.load()
    if x.dim() != 2 or x.shape[1] != 86:
        raise ValueError("MVSNeRF MLP input is (npts, 63 + 20 + 3)")
    params = [_c(t.detach()) for t in params]
    for t, shp in zip(params, MVS_MLP_SHAPES):
        if tuple(t.shape) != shp:
            raise ValueError(f"Renderer_ours parameter of shape {tuple(t.shape)}, expected {shp}")
    npts = x.shape[0]
    act = torch.empty(lib.bmv_mvs_mlp_train_act_floats(npts), device=x.device, dtype=torch.float32)
    scratch = torch.empty(lib.bmv_mvs_mlp_train_scratch_floats(), device=x.device, dtype=torch.float32)
    out = torch.empty(npts, 4, device=x.device, dtype=torch.float32)
    p = _mvs_mlp_params(params)
    with ktimer.region("mvs_mlp_train_fwd"):
        rc = lib.bmv_mvs_mlp_train_fwd(dptr(_c(x), "x"), C.byref(p), npts, dptr(act), dptr(scratch), dptr(out), stream())
    _lib.check(rc, "mvs_mlp_train_fwd")
    return out, act, scratch


def mvs_mlp_train_bwd(params, act, scratch, out, d_out):
    """-> (dx (npts,86), [22 parameter gradients]).  Overwrites `act`."""
    lib = _lib.load()
    params = [_c(t.detach()) for t in params]
    npts = out.shape[0]
    dx = torch.empty(npts, 86, device=out.device, dtype=torch.float32)
    grads = [torch.empty_like(t) for t in params]
    p, g = _mvs_mlp_params(params), _mvs_mlp_params(grads)
    with ktimer.region("mvs_mlp_train_bwd"):
        rc = lib.bmv_mvs_mlp_train_bwd(C.byref(p), dptr(act), dptr(scratch), dptr(_c(out), "out"), dptr(_c(d_out), "d_out"),
                                       npts, dptr(dx), C.byref(g), stream())
    _lib.check(rc, "mvs_mlp_train_bwd")
    return dx, grads


def mvs_mlp(x, blob):
    lead = x.shape[:-1]
    if x.shape[-1] != 86:
        raise ValueError("MVSNeRF MLP input is 63 + 20 + 3 = 86 wide")
    npts = x.numel() // 86
    out = torch.empty(*lead, 4, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with ktimer.region("mvs_mlp"):
        rc = lib.bmv_mvs_mlp_fwd(dptr(_c(x), "x"), dptr(blob, "blob"), npts, dptr(out), stream())
    _lib.check(rc, "mvs_mlp")
    return out


def mvs_render(rays, volume, src_inps, src_exts, src_ixts, near_far, blob, Ns, pad, want_mask=False,
               want_inputs=False, ray_range=None, outs=None):
    """rays (N,8); volume (8,D,hp,wp); src_inps (S,3,H,W); -> raw (N,Ns,4), z (N,Ns), mask (N,Ns)|None,
    inputs86 (N,Ns,86)|None."""
    N = rays.shape[0]
    S, _, H, W = src_inps.shape
    _, D, hp, wp = volume.shape
    dev = rays.device
    if outs is not None:
        raw, z, mask = outs
    else:
        raw = torch.empty(N, Ns, 4, device=dev, dtype=torch.float32) if blob is not None else None
        z = torch.empty(N, Ns, device=dev, dtype=torch.float32)
        mask = torch.empty(N, Ns, device=dev, dtype=torch.float32) if want_mask else None
    x86 = torch.empty(N, Ns, 86, device=dev, dtype=torch.float32) if want_inputs else None
    held = [_c(t) for t in (rays, volume, src_inps, src_exts, src_ixts, near_far)]
    a = _lib.MvsRenderArgs()
    for name, t in zip(("rays", "volume", "src_inps", "src_exts", "src_ixts", "near_far"), held):
        setattr(a, name, dptr(t, name))
    a.blob = dptr(blob, "blob")
    a.N, a.Ns, a.S, a.D, a.hp, a.wp, a.H, a.W, a.pad = N, int(Ns), S, D, hp, wp, H, W, int(pad)
    a.ray_begin, a.ray_end = ray_range if ray_range is not None else (0, N)
    a.raw, a.z_vals, a.mask, a.inputs86 = dptr(raw), dptr(z), dptr(mask), dptr(x86)
    lib = _lib.load()
    with ktimer.region(f"mvs_render[Ns={Ns}]"):
        rc = lib.bmv_mvs_render_fwd(C.byref(a), stream())
    _lib.check(rc, "mvs_render")
    return raw, z, mask, x86


def mvs_march_mask(rays, src_exts, src_ixts, Ns, inv_w, inv_h):
    N = rays.shape[0]
    V = src_exts.shape[0]
    z = torch.empty(N, Ns, device=rays.device, dtype=torch.float32)
    mask = torch.empty(N, Ns, device=rays.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_mvs_march_mask(dptr(_c(rays), "rays"), dptr(_c(src_exts), "src_exts"),
                                      dptr(_c(src_ixts), "src_ixts"), N, int(Ns), V, float(inv_w), float(inv_h),
                                      dptr(z), dptr(mask), stream()), "mvs_march_mask")
    return z, mask


# ======================================================================= backward kernels (fine-tuning)
def composite_bwd(raw, z_vals, d_rgb, d_depth=None):
    Ns = raw.shape[-2]
    nrays = raw.numel() // (Ns * 4)
    d_raw = torch.empty_like(raw, memory_format=torch.contiguous_format)
    lib = _lib.load()
    _lib.check(lib.bmv_composite_bwd(dptr(_c(raw), "raw"), dptr(_c(z_vals), "z_vals"), dptr(_c(d_rgb), "d_rgb"),
                                     dptr(_c(d_depth), "d_depth") if d_depth is not None else None, nrays, Ns,
                                     dptr(d_raw), stream()), "composite_bwd")
    return d_raw


def blend_bwd(raws, masks, d_rgb):
    B, K, N, Ns = raws.shape[:4]
    d_raws = torch.empty_like(raws, memory_format=torch.contiguous_format)
    lib = _lib.load()
    _lib.check(lib.bmv_blend_bwd(dptr(_c(raws), "raws"), dptr(_c(masks).reshape(B, K, N, Ns), "masks"),
                                 dptr(_c(d_rgb), "d_rgb"), B, K, N, Ns, dptr(d_raws), stream()), "blend_bwd")
    return d_raws


def deterministic():
    """bmv_tuning "BMV_DETERMINISTIC" (include/bmv.h; `_lib.set_tuning("BMV_DETERMINISTIC", 1)` or the environment variable
    of that name): the scatter gradients take the *_fixed entry points -- order-independent fixed-point accumulation
    (csrc/scatter.hpp) instead of float atomics -- and the whole backward is bit-reproducible from run to run."""
    return bool(_lib.get_tuning("BMV_DETERMINISTIC"))


def _fixed_ws(n_out, device):
    """Zeroed workspace of a *_fixed call whose scatter outputs hold n_out floats together."""
    words = _lib.load().bmv_fixed_workspace(int(n_out))
    if words < 0:
        _lib.check(int(words), "fixed_workspace")
    return torch.zeros(words, device=device, dtype=torch.int64)


def vox_feat_bwd(uvd01, volume, d_out, ray_w=0, Ns=0):
    """ray_w / Ns: layout hint -- the P samples are Ns per ray, rays row-major over an image ray_w wide (0 = unknown)."""
    B, P = uvd01.shape[:2]
    _, C_, D, h, w = volume.shape
    d_d = torch.empty(B, P, device=volume.device, dtype=torch.float32)
    lib = _lib.load()
    if P == 0:          # an empty ray shard / chunk: no sample, zero gradient (an empty tensor has no device pointer)
        return torch.zeros_like(volume, memory_format=torch.contiguous_format), d_d
    if deterministic():
        d_vol = torch.empty_like(volume, memory_format=torch.contiguous_format)
        ws = _fixed_ws(d_vol.numel(), volume.device)
        _lib.check(lib.bmv_vox_feat_bwd_fixed(dptr(_c(uvd01), "uvd"), dptr(_c(volume), "volume"), dptr(_c(d_out), "d_out"),
                                              B, P, C_, D, h, w, int(ray_w) if Ns else 0, int(Ns), dptr(d_vol), dptr(d_d),
                                              dptr(ws, "workspace", torch.int64), stream()), "vox_feat_bwd_fixed")
        return d_vol, d_d
    d_vol = torch.zeros_like(volume, memory_format=torch.contiguous_format)
    _lib.check(lib.bmv_vox_feat_bwd(dptr(_c(uvd01), "uvd"), dptr(_c(volume), "volume"), dptr(_c(d_out), "d_out"), B, P,
                                    C_, D, h, w, int(ray_w) if Ns else 0, int(Ns), dptr(d_vol), dptr(d_d), stream()),
               "vox_feat_bwd")
    return d_vol, d_d


def img_feat_bwd(xyz, img_feat_rgb, src_exts, src_ixts, tar_ext, render_scale, d_out, n_grad=None, ray_w=0):
    """n_grad: leading channels of img_feat_rgb that need a gradient (default all); ray_w: layout hint -- xyz is
    (B, rays, Ns, 3) with the rays row-major over an image ray_w wide (0 = unknown).  See include/bmv.h."""
    B, S, C_, H, W = img_feat_rgb.shape
    Ns = xyz.shape[-2] if (ray_w and xyz.dim() == 4) else 0
    pts = _c(xyz).reshape(B, -1, 3)
    P = pts.shape[1]
    d_xyz = torch.empty(B, P, 3, device=xyz.device, dtype=torch.float32)
    lib = _lib.load()
    if P == 0:
        return torch.zeros_like(img_feat_rgb, memory_format=torch.contiguous_format), d_xyz.reshape(xyz.shape)
    if deterministic():
        d_img = torch.empty_like(img_feat_rgb, memory_format=torch.contiguous_format)
        ws = _fixed_ws(d_img.numel(), xyz.device)
        _lib.check(lib.bmv_img_feat_bwd_fixed(dptr(pts, "xyz"), dptr(_c(img_feat_rgb), "img"), dptr(_c(src_exts), "src_exts"),
                                              dptr(_c(src_ixts), "src_ixts"), dptr(_c(tar_ext), "tar_ext"),
                                              float(render_scale), dptr(_c(d_out), "d_out"), B, P, S, C_,
                                              C_ if n_grad is None else int(n_grad), H, W, int(ray_w) if Ns else 0, int(Ns),
                                              dptr(d_img), dptr(d_xyz), dptr(ws, "workspace", torch.int64), stream()),
                   "img_feat_bwd_fixed")
        return d_img, d_xyz.reshape(xyz.shape)
    d_img = torch.zeros_like(img_feat_rgb, memory_format=torch.contiguous_format)
    _lib.check(lib.bmv_img_feat_bwd(dptr(pts, "xyz"), dptr(_c(img_feat_rgb), "img"), dptr(_c(src_exts), "src_exts"),
                                    dptr(_c(src_ixts), "src_ixts"), dptr(_c(tar_ext), "tar_ext"), float(render_scale),
                                    dptr(_c(d_out), "d_out"), B, P, S, C_, C_ if n_grad is None else int(n_grad), H, W,
                                    int(ray_w) if Ns else 0, int(Ns), dptr(d_img), dptr(d_xyz), stream()),
               "img_feat_bwd")
    return d_img, d_xyz.reshape(xyz.shape)


def sample_along_depth_bwd(rays12, d_xyz, d_dn, Ns, depth_inv):
    B, N = rays12.shape[:2]
    d_nf = torch.empty(B, N, 2, device=rays12.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.bmv_sample_along_depth_bwd(dptr(_c(rays12), "rays"), dptr(_c(d_xyz), "d_xyz"), dptr(_c(d_dn), "d_dn"),
                                              B, N, int(Ns), int(bool(depth_inv)), dptr(d_nf), stream()),
               "sample_along_depth_bwd")
    return d_nf


def build_rays_bwd(rays, depth, std, near_far, d_nf, Hr, Wr, depth_inv):
    B, N = rays.shape[:2]
    hv, wv = depth.shape[-2:]
    lib = _lib.load()
    if N == 0:
        return (torch.zeros_like(depth, memory_format=torch.contiguous_format),
                torch.zeros_like(std, memory_format=torch.contiguous_format))
    if deterministic():
        d_depth = torch.empty_like(depth, memory_format=torch.contiguous_format)
        d_std = torch.empty_like(std, memory_format=torch.contiguous_format)
        ws = _fixed_ws(2 * d_depth.numel(), depth.device)
        _lib.check(lib.bmv_build_rays_bwd_fixed(dptr(_c(rays), "rays"), dptr(_c(depth), "depth"), dptr(_c(std), "std"),
                                                dptr(_c(near_far), "near_far"), dptr(_c(d_nf), "d_nf"), B, N, hv, wv,
                                                int(Hr), int(Wr), int(bool(depth_inv)), dptr(d_depth), dptr(d_std),
                                                dptr(ws, "workspace", torch.int64), stream()), "build_rays_bwd_fixed")
        return d_depth, d_std
    d_depth = torch.zeros_like(depth, memory_format=torch.contiguous_format)
    d_std = torch.zeros_like(std, memory_format=torch.contiguous_format)
    _lib.check(lib.bmv_build_rays_bwd(dptr(_c(rays), "rays"), dptr(_c(depth), "depth"), dptr(_c(std), "std"),
                                      dptr(_c(near_far), "near_far"), dptr(_c(d_nf), "d_nf"), B, N, hv, wv, int(Hr),
                                      int(Wr), int(bool(depth_inv)), dptr(d_depth), dptr(d_std), stream()),
               "build_rays_bwd")
    return d_depth, d_std


def depth_regress_bwd(depth_prob, depth_values, d_depth, d_std, depth_inv):
    B, D, h, w = depth_values.shape
    d_prob = torch.empty_like(depth_values)
    d_vals = torch.empty_like(depth_values)
    lib = _lib.load()
    _lib.check(lib.bmv_depth_regress_bwd(dptr(_c(depth_prob), "prob"), dptr(_c(depth_values), "values"),
                                         dptr(_c(d_depth), "d_depth"), dptr(_c(d_std), "d_std"), B, D, h, w,
                                         int(bool(depth_inv)), dptr(d_prob), dptr(d_vals), stream()),
               "depth_regress_bwd")
    return d_prob, d_vals


def depth_values_cascade_bwd(depth, std, near_far, d_dv):
    B, h0, w0 = depth.shape
    _, D, h, w = d_dv.shape
    lib = _lib.load()
    if deterministic():
        d_depth = torch.empty_like(depth, memory_format=torch.contiguous_format)
        d_std = torch.empty_like(std, memory_format=torch.contiguous_format)
        ws = _fixed_ws(2 * d_depth.numel(), depth.device)
        _lib.check(lib.bmv_depth_values_cascade_bwd_fixed(dptr(_c(depth), "depth"), dptr(_c(std), "std"),
                                                          dptr(_c(near_far), "near_far"), dptr(_c(d_dv), "d_dv"), B, h0, w0,
                                                          h, w, D, dptr(d_depth), dptr(d_std),
                                                          dptr(ws, "workspace", torch.int64), stream()),
                   "depth_values_cascade_bwd_fixed")
        return d_depth, d_std
    d_depth = torch.zeros_like(depth, memory_format=torch.contiguous_format)
    d_std = torch.zeros_like(std, memory_format=torch.contiguous_format)
    _lib.check(lib.bmv_depth_values_cascade_bwd(dptr(_c(depth), "depth"), dptr(_c(std), "std"),
                                                dptr(_c(near_far), "near_far"), dptr(_c(d_dv), "d_dv"), B, h0, w0, h, w,
                                                D, dptr(d_depth), dptr(d_std), stream()), "depth_values_cascade_bwd")
    return d_depth, d_std


def sweep_variance_bwd(feats, proj, depth_values, d_var, want_depth_grad, algo=None):
    """d variance -> (d feats (B,S,C,Hs,Ws), d depth_values or None).  algo None / "cl": for S = 3 and C in {16, 32} the
    channel-last kernel (csrc/sweep_bwd_cl.hip: the gradient is accumulated channel-last with the channel on the lane
    and handed back as a (B,S,C,Hs,Ws) VIEW of that buffer); "planar" (or BMV_SWEEP_BWD=planar): the LDS-window kernel
    on the reference layout."""
    B, S, C_, Hs, Ws = feats.shape
    _, D, h, w = depth_values.shape
    lib = _lib.load()
    if deterministic():          # bit-reproducible: the planar kernel with fixed-point accumulation (csrc/scatter.hpp)
        d_feats = torch.empty_like(feats, memory_format=torch.contiguous_format)
        d_dv = torch.empty_like(depth_values, memory_format=torch.contiguous_format) if want_depth_grad else None
        ws = _fixed_ws(d_feats.numel() + (d_dv.numel() if want_depth_grad else 0), feats.device)
        _lib.check(lib.bmv_sweep_variance_bwd_fixed(dptr(_c(feats), "feats"), dptr(_c(proj), "proj"),
                                                    dptr(_c(depth_values), "depth_values"), dptr(_c(d_var), "d_var"), B, S,
                                                    C_, Hs, Ws, D, h, w, dptr(d_feats), dptr(d_dv),
                                                    dptr(ws, "workspace", torch.int64), stream()), "sweep_variance_bwd_fixed")
        return d_feats, d_dv
    if algo is None:
        algo = switches.get("BMV_SWEEP_BWD")
    if algo == "cl" and S == 3 and C_ in (16, 32):
        cl = feats.permute(0, 1, 3, 4, 2)
        cl = cl if cl.is_contiguous() else nchw_to_nhwc(feats)
        d_cl = torch.zeros(B, S, Hs, Ws, C_, device=feats.device, dtype=torch.float32)
        d_dv = torch.empty_like(depth_values, memory_format=torch.contiguous_format) if want_depth_grad else None
        _lib.check(lib.bmv_sweep_variance_bwd_cl(dptr(cl, "feats_cl"), dptr(_c(proj), "proj"),
                                                 dptr(_c(depth_values), "depth_values"), dptr(_c(d_var), "d_var"), B, S, C_,
                                                 Hs, Ws, D, h, w, dptr(d_cl), dptr(d_dv), stream()), "sweep_variance_bwd_cl")
        return d_cl.permute(0, 1, 4, 2, 3), d_dv
    d_feats = torch.zeros_like(feats, memory_format=torch.contiguous_format)
    d_dv = torch.zeros_like(depth_values) if want_depth_grad else None
    _lib.check(lib.bmv_sweep_variance_bwd(dptr(_c(feats), "feats"), dptr(_c(proj), "proj"),
                                          dptr(_c(depth_values), "depth_values"), dptr(_c(d_var), "d_var"), B, S, C_,
                                          Hs, Ws, D, h, w, dptr(d_feats), dptr(d_dv), stream()), "sweep_variance_bwd")
    return d_feats, d_dv


def nerf_pack_bwd_weights(tensors, feat_ch):
    lib = _lib.load()
    n = lib.bmv_nerf_bwd_blob_size(int(feat_ch))
    if n < 0:
        _lib.check(n, "nerf_bwd_blob_size")
    out = torch.empty(n, device=tensors[0].device, dtype=torch.float32)
    held = [_c(t.detach()) for t in tensors]
    params = _lib.NerfParams(*[dptr(t, f"nerf param {i}") for i, t in enumerate(held)])
    _lib.check(lib.bmv_nerf_pack_bwd_weights(C.byref(params), int(feat_ch), dptr(out), stream()), "nerf_pack_bwd_weights")
    return out


def nerf_bwd_rows(feat_ch, S=3):
    lib = _lib.load()
    ir = C.c_int(0)
    r = lib.bmv_nerf_bwd_rows(int(feat_ch), int(S), C.byref(ir))
    if r < 0:
        _lib.check(r, "nerf_bwd_rows")
    return r, ir.value


def nerf_param_shapes(feat_ch):
    """Shapes of the 16 parameter tensors in NERF_PARAM_ORDER (weight, bias)."""
    F = int(feat_ch) + 3
    return [(F, 4), (F,), (32, 3 * F), (32,), (1, 32), (1,), (16, 32), (16,), (64, 24), (64,), (1, 64), (1,),
            (64, 88 + F + 4), (64,), (1, 64), (1,)]


def nerf_mlp_bwd(vox_feat_t, img_feat_rgb_dir, d_out, blob_fwd, blob_bwd, feat_ch):
    """-> d_vox (8,P), d_img (S,IR,P), grads: the 16 parameter gradients in NERF_PARAM_ORDER (weight, bias); S = the
    source views of img_feat_rgb_dir (P, S, feat_ch + 7), 2..4.
    Data path, weight gradients (MFMA over the sample dimension) and their reduction are three launches inside
    bmv_nerf_mlp_bwd; the per-tile matrices between them live in a workspace sized by bmv_nerf_bwd_workspace."""
    npts = vox_feat_t.numel() // 8
    S = int(img_feat_rgb_dir.shape[-2])
    _, IR = nerf_bwd_rows(feat_ch, S)
    dev = vox_feat_t.device
    lib = _lib.load()
    n_ws = lib.bmv_nerf_bwd_workspace(int(feat_ch), S, npts)
    if n_ws < 0:
        _lib.check(int(n_ws), "nerf_bwd_workspace")
    ws = torch.empty(n_ws, device=dev, dtype=torch.float32)
    d_vox = torch.empty(8, npts, device=dev, dtype=torch.float32)
    d_img = torch.zeros(S, IR, npts, device=dev, dtype=torch.float32)
    grads = [torch.empty(s, device=dev, dtype=torch.float32) for s in nerf_param_shapes(feat_ch)]
    gp = _lib.NerfParams(*[dptr(t, f"nerf grad {i}") for i, t in enumerate(grads)])
    with ktimer.region(f"nerf_mlp_bwd[feat={feat_ch}]"):
        rc = lib.bmv_nerf_mlp_bwd(dptr(_c(vox_feat_t), "vox_feat"), dptr(_c(img_feat_rgb_dir), "img"),
                                  dptr(_c(d_out), "d_out"), dptr(blob_fwd, "blob_fwd"), dptr(blob_bwd, "blob_bwd"),
                                  int(feat_ch), S, npts, dptr(ws), dptr(d_vox), dptr(d_img), C.byref(gp), stream())
    _lib.check(rc, "nerf_mlp_bwd")
    return d_vox, d_img, grads


def conv_wgrad(big, small, stride, kd, k):
    """big (B,Cb,[Db,]Hb,Wb) zero-padded input side, small (B,Cs,[Ds,]Hs,Ws) output side -> G (Cs,Cb,[kd,]k,k) with
    G[s,b,taps] = sum_n sum_p small[n,s,p] * big[n,b, stride*p + tap] (include/bmv.h: bmv_conv_wgrad)."""
    lib = _lib.load()
    is3d = small.dim() == 5
    B, Cb = big.shape[:2]
    Cs = small.shape[1]
    Db = big.shape[2] if is3d else 1
    Ds = small.shape[2] if is3d else 1
    Hb, Wb = big.shape[-2:]
    Hs, Ws = small.shape[-2:]
    n_ws = lib.bmv_conv_wgrad_workspace(B, Cs, Cb, Ds, Hs, Ws, kd, k)
    if n_ws < 0:
        _lib.check(int(n_ws), "conv_wgrad_workspace")
    ws = torch.empty(n_ws, device=small.device, dtype=torch.float32)
    G = torch.empty((Cs, Cb, kd, k, k) if is3d else (Cs, Cb, k, k), device=small.device, dtype=torch.float32)
    with ktimer.region("conv_wgrad"):
        rc = lib.bmv_conv_wgrad(dptr(_c(big), "big"), dptr(_c(small), "small"), B, Cb, Db, Hb, Wb, Cs, Ds, Hs, Ws, int(kd), int(k),
                                int(stride), dptr(ws), dptr(G), stream())
    _lib.check(rc, "conv_wgrad")
    return G


def _act_slope(relu):
    """relu: False -> 1 (no activation), True -> 0 (ReLU), a float -> that leaky slope."""
    return 1.0 if relu is False else 0.0 if relu is True else float(relu)


def bn_train_fwd(x, weight, bias, running_mean, running_var, eps, momentum, relu):
    """Training-mode batch norm (+ ReLU / leaky ReLU: `relu` = True / a slope) of x (N,C,*spatial); updates the running
    statistics in place.  -> y, save_mean (C), save_invstd (C)."""
    lib = _lib.load()
    x = _c(x)
    N, C_ = x.shape[:2]
    S = x.numel() // (N * C_)
    chunks = lib.bmv_bn_chunks(N, S)
    ws = torch.empty(C_ * chunks * 3, device=x.device, dtype=torch.float32)
    mean = torch.empty(C_, device=x.device, dtype=torch.float32)
    invstd = torch.empty(C_, device=x.device, dtype=torch.float32)
    y = torch.empty_like(x)
    with ktimer.region("bn_train_fwd"):
        rc = lib.bmv_bn_train_fwd(dptr(x, "x"), dptr(weight, "weight"), dptr(bias, "bias"), dptr(running_mean, "running_mean"),
                                  dptr(running_var, "running_var"), N, C_, S, float(eps), float(momentum), _act_slope(relu),
                                  dptr(ws), dptr(mean), dptr(invstd), dptr(y), stream())
    _lib.check(rc, "bn_train_fwd")
    return y, mean, invstd


def bn_train_bwd(x, y, dy, weight, mean, invstd, relu):
    """-> dx, dweight (C), dbias (C)."""
    lib = _lib.load()
    x, dy = _c(x), _c(dy)
    N, C_ = x.shape[:2]
    S = x.numel() // (N * C_)
    chunks = lib.bmv_bn_chunks(N, S)
    ws = torch.empty(C_ * chunks * 2 + 2 * C_, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    dw = torch.empty(C_, device=x.device, dtype=torch.float32)
    db = torch.empty(C_, device=x.device, dtype=torch.float32)
    with ktimer.region("bn_train_bwd"):
        rc = lib.bmv_bn_train_bwd(dptr(x, "x"), dptr(_c(y), "y") if relu is not False else None, dptr(dy, "dy"), dptr(weight, "weight"),
                                  dptr(mean), dptr(invstd), N, C_, S, _act_slope(relu), dptr(ws), dptr(dx), dptr(dw), dptr(db),
                                  stream())
    _lib.check(rc, "bn_train_bwd")
    return dx, dw, db


def mvs_sweep_bwd(feats, proj, depth_values, d_volume, pad):
    """d_volume (B, 3S+C, D, hp, wp) -> d_feats (B,S,C,h,w)."""
    B, S, C_, h, w = feats.shape
    D = depth_values.shape[1]
    lib = _lib.load()
    if deterministic():
        d_feats = torch.empty_like(feats, memory_format=torch.contiguous_format)
        ws = _fixed_ws(d_feats.numel(), feats.device)
        _lib.check(lib.bmv_mvs_sweep_bwd_fixed(dptr(_c(feats), "feats"), dptr(_c(proj), "proj"),
                                               dptr(_c(depth_values), "depth_values"), dptr(_c(d_volume), "d_volume"), B, S,
                                               C_, h, w, D, int(pad), dptr(d_feats), dptr(ws, "workspace", torch.int64),
                                               stream()), "mvs_sweep_bwd_fixed")
        return d_feats
    d_feats = torch.zeros_like(feats, memory_format=torch.contiguous_format)
    _lib.check(lib.bmv_mvs_sweep_bwd(dptr(_c(feats), "feats"), dptr(_c(proj), "proj"), dptr(_c(depth_values), "depth_values"),
                                     dptr(_c(d_volume), "d_volume"), B, S, C_, h, w, D, int(pad), dptr(d_feats), stream()),
               "mvs_sweep_bwd")
    return d_feats


def mvs_vol_feat_bwd(rays, src_ext0, src_ixt0, near_far, d_feat, H, W, vol_shape, pad):
    """d_feat (N,Ns,8) -> d_volume (8,D,hp,wp)."""
    N, Ns = d_feat.shape[:2]
    _, D, hp, wp = vol_shape
    lib = _lib.load()
    if N == 0:
        return torch.zeros(8, D, hp, wp, device=d_feat.device, dtype=torch.float32)
    if deterministic():
        d_vol = torch.empty(8, D, hp, wp, device=d_feat.device, dtype=torch.float32)
        ws = _fixed_ws(d_vol.numel(), d_feat.device)
        _lib.check(lib.bmv_mvs_vol_feat_bwd_fixed(dptr(_c(rays), "rays"), dptr(_c(src_ext0), "src_ext0"),
                                                  dptr(_c(src_ixt0), "src_ixt0"), dptr(_c(near_far), "near_far"),
                                                  dptr(_c(d_feat), "d_feat"), N, int(Ns), int(H), int(W), D, hp, wp, int(pad),
                                                  dptr(d_vol), dptr(ws, "workspace", torch.int64), stream()),
                   "mvs_vol_feat_bwd_fixed")
        return d_vol
    d_vol = torch.zeros(8, D, hp, wp, device=d_feat.device, dtype=torch.float32)
    _lib.check(lib.bmv_mvs_vol_feat_bwd(dptr(_c(rays), "rays"), dptr(_c(src_ext0), "src_ext0"), dptr(_c(src_ixt0), "src_ixt0"),
                                        dptr(_c(near_far), "near_far"), dptr(_c(d_feat), "d_feat"), N, int(Ns), int(H), int(W), D,
                                        hp, wp, int(pad), dptr(d_vol), stream()), "mvs_vol_feat_bwd")
    return d_vol
