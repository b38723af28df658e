"""ENeRF on the MI355X hot-path kernels.

Same module boundary as the reference's lib/networks/enerf/network.py:11-113:
`Network()` is an nn.Module with sub-modules feature_net, cost_reg_{i}, nerf_{i}
(identical state-dict keys), `forward(batch)` returns
{rgb,depth,weights,depth_mvs,std}_level{i} for every level with render_if[i].

Inside, a cascade level is five launches instead of the reference's ~150 torch
ops: projection matrices -> depth hypotheses -> fused plane-sweep variance ->
[3-D regulariser, csrc/conv.hip MFMA engine] -> depth regression -> ONE fused kernel from
rays to composited pixels (per-ray bounds, samples, volume + image lookups,
MFMA MLP, alpha compositing).  No warped volume, sample tensor or per-view
feature tensor is ever materialised.
"""

import torch
import torch.nn as nn

from ... import autograd as A
from ... import convnet, ops, switches
from ...config import cfg
from .cnn import CostRegNet, FeatureNet, MinCostRegNet, engine_ok
from .nerf import NeRF


def _views(t, B, V, h, w):
    """(B V, C, h, w) feature maps -> (B, V, C, h, w); ops.QuadFeats (the sweep's layout) keep their wrapper."""
    return t.reshape_views(B, V) if isinstance(t, ops.QuadFeats) else t.reshape(B, V, -1, h, w)


class LevelState:
    """What one cascade level hands to the next (and to the renderer)."""
    __slots__ = ("depth", "std", "near_far", "feature_volume", "depth_values", "keep")

    def __init__(self):
        self.depth = self.std = self.near_far = self.feature_volume = self.depth_values = None
        self.keep = None      # tensors a side stream still reads: kept alive until the level state dies


def _check_renderer_case(S, Ns, feat_ch, depth_inv):
    """The fused renderer (bmv_render_rays_fwd) is instantiated for S in {2, 3, 4} source views x (feat_ch 8, linear
    depth, Ns in {1, 2, 4, 8}) and (feat_ch 32, inverse depth, Ns in {2, 4, 8}): the cascade levels of every shipped
    config and their `num_samples` / `test_input_views` variations.  Anything else is refused HERE, by name, before a
    launch (the reference's torch ops take any combination, lib/networks/enerf/network.py:24-43)."""
    ok = (2 <= S <= 4) and ((feat_ch == 8 and not depth_inv and Ns in (1, 2, 4, 8)) or
                            (feat_ch == 32 and depth_inv and Ns in (2, 4, 8)))
    if not ok:
        raise NotImplementedError(
            f"fused renderer: no kernel for {S} source views, {Ns} samples per ray, {feat_ch} feature channels, "
            f"depth_inv={depth_inv}.  Built: S in (2, 3, 4) with (feat_ch 8, depth_inv False, Ns in 1/2/4/8) or "
            "(feat_ch 32, depth_inv True, Ns in 2/4/8) -- INTEGRATION.md, 'Limits'")


class Network(nn.Module):
    def __init__(self):
        super().__init__()
        cc = cfg.enerf.cas_config
        self.feature_net = FeatureNet()
        for i in range(cc.num):
            width = int(32 * 2 ** (-i))
            setattr(self, f"cost_reg_{i}", MinCostRegNet(width) if i == 0 else CostRegNet(width))
            setattr(self, f"nerf_{i}", NeRF(feat_ch=cc.nerf_model_feat_ch[i] + 3))
        # optional intra-frame ray sharding (boostmvsnerfs_amd/sharding.py): render rays [begin, end) only
        self.ray_range = None
        self.sweep_algo = 0
        # inference: run the level-0 cascade chain on a second stream under FeatureNet's top-down path.
        # Off by default: measured +1.2 % frames/s under graph replay (the concurrent kernels slow each other) and it
        # takes the source features of the level-1 sweep out of L2 (that kernel: 29.1 -> 32.7 us).
        # 0: one stream; 1: the whole level-0 chain on a side stream under FeatureNet's top-down path; 2: only what follows
        # the level-0 sweep (regulariser + depth regression) -- the sweep stays on the main stream.  By default the fork
        # is used under HIP-graph capture only (replay: +3 % frames/s); issued eagerly a 512x640 frame is bound by the
        # host's ~45 launches and the extra stream traffic costs 3 % (BMV_OVERLAP_EAGER=1 forks there too).
        self.overlap_front = switches.get("BMV_OVERLAP")
        self.lookup_records = switches.on("BMV_LOOKUP_RECORDS")
        self.frame_setup = switches.on("BMV_FRAME_SETUP")
        self.depth_maps_through_table = switches.on("BMV_DEPTH_MAPS_TABLE")
        self._pre = None
        self.volume_records = switches.on("BMV_VOLUME_RECORDS")
        self.overlap_eager = switches.on("BMV_OVERLAP_EAGER")
        # the renderer's full-resolution feature map (FeatureNet's last launch) beside the level-1 chain instead of inside
        # the level-0 window (round 6: that window is bound by the level-0 regulariser since csrc/fpn_s.hip)
        self.defer_f0 = switches.on("BMV_DEFER_F0")
        self._f0_pending = None
        from ...autograph import AutoGraph
        object.__setattr__(self, "_autograph", AutoGraph(self))
        # autograph opt-ins (autograph.py): the caller declares its batch resident / accepts outputs that the next
        # forward overwrites.  Default: forward neither writes its inputs nor hands out memory it will reuse
        self.resident_inputs = False
        self.alias_outputs = False
        # load_state_dict (assign=True replaces the Parameter objects) makes captured frames stale
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._autograph.invalidate())
        self._side_stream = None

    # ------------------------------------------------------------------ 2-D features
    def forward_feat(self, x):
        """(B,V,3,H,W) -> {'level_0': 32ch @ 1/4, 'level_1': 16ch @ 1/2, 'level_2': 8ch @ 1}
        (lib/networks/enerf/network.py:58-67)."""
        B, V, C, H, W = x.shape
        self.feature_net.quad_out = self.sweep_algo == 0 or self.sweep_algo >= 500
        coarse, mid, fine = self.feature_net(x.reshape(B * V, C, H, W))
        return {"level_0": _views(coarse, B, V, H // 4, W // 4),
                "level_1": _views(mid, B, V, H // 2, W // 2),
                "level_2": fine.reshape_views(B, V) if isinstance(fine, convnet.LookupRecords) else fine.reshape(B, V, -1, H, W)}

    def set_volume_records(self, on):
        """With image records on, the regulariser of every level rendered from them writes its feature volume as voxel
        records too (convnet.VolumeRecords: the renderer's record kernels take both)."""
        cc = cfg.enerf.cas_config
        for i in range(cc.num):
            getattr(self, f"cost_reg_{i}").volume_records = bool(on and self.volume_records and cc.render_if[i]
                                                                 and cc.render_im_feat_level[i] == 2)

    def wants_lookup_records(self):
        """The full-resolution feature map can leave FeatureNet as the fused renderer's lookup records (one 48-byte
        record per pixel: 8 feature channels + the source colours) when nothing but that kernel reads it: inference,
        and every rendered level that takes its image features from level 2 does so at render scale 1."""
        cc = cfg.enerf.cas_config
        if not self.lookup_records or self.wants_grad():
            return False
        users = [i for i in range(cc.num) if cc.render_if[i] and cc.render_im_feat_level[i] == 2]
        # ... and: level 2 is never SWEPT as cost-volume features (a third cascade level would hand the records to
        # the plane sweep), and the renderer has a record kernel for the level's sample count
        if cc.num > 2:
            return False
        return bool(users) and all(cc.render_scale[i] == 1.0 and cc.im_ibr_scale[i] == 1.0
                                   and getattr(self, f"nerf_{i}").feat_ch - 3 == 8
                                   and int(cc.num_samples[i]) in (1, 2, 4, 8) for i in users)

    # ------------------------------------------------------------------ cost volume of one level
    def camera_only(self, views, batch):
        """What a frame needs from the cameras alone -- the projection matrices of every cascade level and level 0's
        uniform hypotheses -- as ONE launch (ops.frame_setup) instead of three in front of the sweeps (inference)."""
        cc = cfg.enerf.cas_config
        src_inps, src_exts, src_ixts = views
        H, W = src_inps.shape[-2:]
        proj, dv0 = ops.frame_setup(src_exts, src_ixts, batch["tar_ext"], batch["tar_ixt"],
                                    [cc.im_feat_scale[i] for i in range(cc.num)], [cc.volume_scale[i] for i in range(cc.num)],
                                    batch["near_far"], cc.volume_planes[0], int(H * cc.volume_scale[0]),
                                    int(W * cc.volume_scale[0]), cc.depth_inv[0])
        return {"proj": proj, "dv0": dv0}

    def level_front(self, i, feats_i, views, batch, prev, view_ids=None, fork_after_sweep=None, pre=None):
        """Plane sweep + regulariser + depth regression (network.py:81-90).
        `fork_after_sweep`: a side stream that takes everything AFTER the sweep (regulariser, depth regression); the
        caller joins it.  The sweep itself stays on the current stream.
        `views` = (src_inps, src_exts, src_ixts) of the S views of this cost volume.  With `view_ids` (B,S) int32
        (inference), `feats_i` and `src_inps` hold ALL source views and the kernels pick the volume's views by index
        (no gathered copies); src_exts / src_ixts are the S picked ones either way."""
        cc = cfg.enerf.cas_config
        src_inps, src_exts, src_ixts = views
        H, W = src_inps.shape[-2:]
        h, w = int(H * cc.volume_scale[i]), int(W * cc.volume_scale[i])
        D = cc.volume_planes[i]
        st = LevelState()
        train = self.wants_grad()             # fine-tuning: autograd Functions (HIP forward + HIP backward)
        if (prev is None or prev.depth is None) and pre is not None and i == 0:
            st.depth_values, st.near_far = pre["dv0"]
        elif prev is None or prev.depth is None:
            st.depth_values, st.near_far = ops.depth_values_uniform(batch["near_far"], D, h, w, cc.depth_inv[i])
        else:
            if not cc.depth_inv[i - 1] or cc.depth_inv[i]:
                raise NotImplementedError("cascade levels must go disparity -> depth")
            if train:
                st.depth_values, st.near_far = A.DepthValuesCascade.apply(prev.depth, prev.std, prev.near_far, h, w, D)
            else:
                st.depth_values, st.near_far = ops.depth_values_cascade(prev.depth, prev.std, prev.near_far, h, w, D)
        proj = pre["proj"][i] if pre is not None else ops.proj_mats(
            src_exts, src_ixts, batch["tar_ext"], batch["tar_ixt"], cc.im_feat_scale[i], cc.volume_scale[i])
        uniform = prev is None or prev.depth is None     # level 0: one hypothesis per plane (enerf/utils.py:104-111)
        if train:
            variance = A.SweepVariance.apply(feats_i, proj, st.depth_values, self.sweep_algo)
        elif view_ids is not None:
            variance = ops.sweep_variance_views(feats_i, view_ids, proj, st.depth_values, plane_uniform=uniform,
                                                quad_out=getattr(self, f"cost_reg_{i}").takes_quad_volume())
        else:
            variance = ops.sweep_variance(feats_i, proj, st.depth_values, algo=self.sweep_algo, plane_uniform=uniform,
                                          quad_out=getattr(self, f"cost_reg_{i}").takes_quad_volume())
        if fork_after_sweep is not None:
            fork_after_sweep.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(fork_after_sweep):
                st.feature_volume, depth_prob = getattr(self, f"cost_reg_{i}")(variance)
                st.depth, st.std = ops.depth_regress(depth_prob, st.depth_values, cc.depth_inv[i])
            # the variance volume was allocated on the current stream and is read on the side one: it must not go back
            # to the allocator (and be handed to the next main-stream allocation) before the caller has joined
            st.keep = (variance, depth_prob)
            return st
        st.feature_volume, depth_prob = getattr(self, f"cost_reg_{i}")(variance)
        if train:
            st.depth, st.std = A.DepthRegress.apply(depth_prob, st.depth_values, cc.depth_inv[i])
        else:
            # (a rendered level's std / depth_mvs maps are frame outputs: written through the pointer table as well)
            outs = (("std",) + (() if cc.depth_inv[i] else ("depth",))) if (cc.render_if[i] and self.depth_maps_through_table) else ()
            st.depth, st.std = ops.depth_regress(depth_prob, st.depth_values, cc.depth_inv[i], frame_outputs=outs)
        return st

    def wants_grad(self):
        """The differentiable (op-by-op) path is taken only when something can receive a gradient: grad mode on AND a
        parameter that requires it.  An eval-mode call that merely forgot torch.no_grad() on a frozen network keeps
        the fused kernels (and with them chunking and ray sharding)."""
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    # ------------------------------------------------------------------ unfused renderer (differentiable)
    def _render_bounded(self, i, rays12, feature_volume, im_feat, views, tar_ext, nerf, mode=0):
        """Bounded rays (B,n,12) -> composited pixels, one HIP kernel per reference op (network.py:24-44): the
        per-sample tensors are materialised between ops as the reference does, so every op has a HIP backward
        (backward contract).  Differentiable when grad mode is on."""
        cc = cfg.enerf.cas_config
        src_inps, src_exts, src_ixts = views
        B = src_inps.shape[0]
        H, W = src_inps.shape[-2:]
        rs = cc.render_scale[i]
        Hr, Wr = ops.scaled_size(H, W, rs)
        up = cc.render_scale[i] / cc.im_ibr_scale[i]
        if up != 1.0:      # network.py:29-32 (no shipped config takes this branch; torch's resize, as there)
            b_, s_, c_, h_, w_ = im_feat.shape
            im_feat = torch.nn.functional.interpolate(im_feat.reshape(b_ * s_, c_, h_, w_), None, scale_factor=up,
                                                      align_corners=True, mode="bilinear")
            im_feat = im_feat.view(b_, s_, c_, int(h_ * up), int(w_ * up))
        Ns, inv = cc.num_samples[i], cc.depth_inv[i]
        if not 2 <= src_exts.shape[1] <= 4:
            raise NotImplementedError(f"{src_exts.shape[1]} source views per cost volume: the fused MLP kernels (forward and "
                                      "backward) are built for 2, 3 or 4 (INTEGRATION.md, 'Limits')")
        xyz, uvd, z = A.SampleAlongDepth.apply(rays12.contiguous(), Ns, inv)
        uvd01 = torch.stack([uvd[..., 0] / (Wr - 1), uvd[..., 1] / (Hr - 1), uvd[..., 2]], -1).reshape(B, -1, 3)
        full = xyz.shape[1] == Hr * Wr          # a whole frame of rays (train_img): the scatter kernels tile it in 2-D
        vox = A.VoxFeat.apply(uvd01, feature_volume, Wr if full else 0, Ns)
        img = torch.cat([im_feat, ops.unpreprocess(src_inps, Hr, Wr)], 2)
        # the colour channels are data; a full frame of rays (train_img) lets the backward tile them in 2-D
        feat = A.ImgFeat.apply(xyz, img, src_exts, src_ixts, tar_ext, rs,
                               None if src_inps.requires_grad else im_feat.shape[2], Wr if full else 0)
        params = [t for lin in nerf._linears() for t in (lin.weight, lin.bias)]
        raw = A.NerfMLP.apply(vox, feat, nerf.feat_ch - 3, *params).reshape(B, -1, Ns, 4)      # S in {2, 3, 4}
        if mode == 1:
            with torch.no_grad():
                mask = ops.mask_viewport(xyz, src_exts, src_ixts, Wr - 1, Hr - 1).view(B, -1, Ns)
            return raw, z, mask
        return A.Composite.apply(raw, z, cfg.enerf.white_bkgd)

    def render_level_train(self, i, st, im_feat, views, batch, mode=0):
        """Training renderer of one level: build_rays (a6) + the unfused chain."""
        cc = cfg.enerf.cas_config
        H, W = views[0].shape[-2:]
        rs = cc.render_scale[i]
        if cc.render_scale[i] / cc.im_ibr_scale[i] != 1.0:
            raise NotImplementedError("im_feat must be at the render resolution (true for every shipped config)")
        rays = batch[f"rays_{i}"]
        if self.ray_range is not None:          # the rays of this rank (sharding.ray_slice), as the fused path
            rays = rays[:, self.ray_range[0]:self.ray_range[1]].contiguous()
        rays12 = A.BuildRays.apply(rays, st.depth, st.std, st.near_far, int(H * rs), int(W * rs), cc.depth_inv[i])
        return self._render_bounded(i, rays12, st.feature_volume, im_feat, views, batch["tar_ext"], getattr(self, f"nerf_{i}"), mode)

    # ------------------------------------------------------------------ fused renderer of one level
    def render_level(self, i, st, im_feat, views, batch, mode=0, outs=None, view_ids=None):
        """rays -> pixels (mode 0) or raw MLP outputs + depths + visibility (mode 1)
        (network.py:24-55, boost_enerf/network.py:123-161).  `view_ids`: see level_front (needs render_scale 1)."""
        if isinstance(im_feat, ops.QuadFeats):     # a cascade level rendered from a map the sweep also reads (level 0 /
            im_feat = im_feat.to_nchw()            # level 1 image features): the renderer takes the planar tensor
        cc = cfg.enerf.cas_config
        src_inps, src_exts, src_ixts = views
        H, W = src_inps.shape[-2:]
        rs = cc.render_scale[i]
        Hr, Wr = ops.scaled_size(H, W, rs)
        if cc.render_scale[i] / cc.im_ibr_scale[i] != 1.0:
            raise NotImplementedError("im_feat must be at the render resolution (true for every shipped config)")
        packed = None
        if isinstance(im_feat, convnet.LookupRecords):
            if rs != 1.0:
                raise ValueError("lookup records carry the source colours at full resolution (render_scale 1)")
            packed, im_feat, rgb_src, affine = im_feat.t, None, None, True
        elif rs == 1.0:
            rgb_src, affine = src_inps, True
        else:
            if view_ids is not None:
                raise ValueError("view_ids need render_scale 1 (the resized colour maps are per cost volume)")
            rgb_src, affine = ops.unpreprocess(src_inps, Hr, Wr), False
        nerf = getattr(self, f"nerf_{i}")
        _check_renderer_case(src_exts.shape[1], int(cc.num_samples[i]), nerf.feat_ch - 3, bool(cc.depth_inv[i]))
        rays = batch[f"rays_{i}"]
        N = rays.shape[1]
        begin, end = self.ray_range if self.ray_range is not None else (0, N)
        chunk = int(cfg.enerf.chunk_size)
        # one launch per chunk keeps the reference's memory bound; with fused kernels a whole frame is one chunk
        for c0 in range(begin, end, chunk):
            o = ops.render_rays(rays, st.depth, st.std, st.near_far, st.feature_volume, im_feat, rgb_src, src_exts,
                                src_ixts, batch["tar_ext"], nerf.packed_weights(), feat_ch=nerf.feat_ch - 3,
                                Ns=cc.num_samples[i], depth_inv=cc.depth_inv[i], Hr=Hr, Wr=Wr, render_scale=rs,
                                rgb_affine=affine, white_bkgd=cfg.enerf.white_bkgd, mode=mode,
                                ray_range=(c0, min(c0 + chunk, end)), outs=outs, view_ids=view_ids, im_packed=packed)
            outs = o      # every chunk writes its own ray slice of the same buffers
        if (begin, end) != (0, N):
            outs = tuple(t[:, begin:end] for t in outs)
        return outs

    # ------------------------------------------------------------------ the reference's per-chunk entry points
    def render_rays(self, rays, **kwargs):
        """network.py:24-44, same keywords: `rays` are bounded rays (B,n,12) from build_rays, `level`, `batch`,
        `im_feat`, `feature_volume`, `nerf_model`.  Returns raw2outputs' dict.  forward() does not come through here
        (it uses the fused kernel of render_level); this is the op-by-op HIP chain, differentiable."""
        level, batch = kwargs["level"], kwargs["batch"]
        views = (batch["src_inps"], batch["src_exts"], batch["src_ixts"])
        rgb, depth, weights = self._render_bounded(level, rays, kwargs["feature_volume"], kwargs["im_feat"], views,
                                                   batch["tar_ext"], kwargs["nerf_model"])
        return {"rgb": rgb, "depth": depth, "weights": weights}

    def batchify_rays(self, rays, **kwargs):
        """network.py:46-56: render_rays over chunks of cfg.enerf.chunk_size rays, concatenated."""
        chunk = int(cfg.enerf.chunk_size)
        parts = [self.render_rays(rays[:, c0:c0 + chunk], **kwargs) for c0 in range(0, rays.shape[1], chunk)]
        return {k: torch.cat([p[k] for p in parts], 1) for k in parts[0]}

    # ------------------------------------------------------------------ overlapped front end (inference)
    def _front_overlapped(self, batch, views):
        """FeatureNet's coarsest map is all the level-0 cost volume needs: the level-0 chain (sweep, 3-D
        regulariser, depth regression: ~20 short launches that fill a fraction of the chip) runs on a second
        HIP stream while the main stream finishes FeatureNet's top-down path (4 large launches).
        Returns (feats, level-0 state); the main stream has waited for the side stream."""
        x = batch["src_inps"]
        B, V, C, H, W = x.shape
        fn = self.feature_net
        self._pre = self.camera_only(views, batch) if self.frame_setup else None
        fn.quad_out = self.sweep_algo == 0 or self.sweep_algo >= 500
        c0, c1, p2, p2_cl = fn.engine_bottom_up(x.reshape(B * V, C, H, W))
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(priority=switches.get("BMV_SIDE_PRIO"))
        side = self._side_stream
        level0 = _views(p2_cl, B, V, H // 4, W // 4)
        if self.overlap_front == 2:
            st0 = self.level_front(0, level0, views, batch, None, fork_after_sweep=side, pre=self._pre)
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                st0 = self.level_front(0, level0, views, batch, None)
        defer = bool(self.defer_f0)
        f1, f0 = fn.engine_top_down(c0, c1, p2, rgb=x.reshape(B * V, C, H, W) if fn.pack_lookup else None, defer_f0=defer)
        main.wait_stream(side)
        if defer:       # the side stream is free again: the full-resolution map runs there under the level-1 chain
            side.wait_stream(main)
            with torch.cuda.stream(side):
                f0 = f0()
            self._f0_pending = side
            if not torch.cuda.is_current_stream_capturing():
                (f0.t if isinstance(f0, convnet.LookupRecords) else f0).record_stream(main)
        if not torch.cuda.is_current_stream_capturing():   # (a graph capture owns its memory pool)
            for name in ("depth", "std", "near_far", "feature_volume", "depth_values"):
                t = getattr(st0, name)           # allocated under the side stream, consumed on the main one
                if isinstance(t, convnet.VolumeRecords):
                    t = t.t
                if t is not None:
                    t.record_stream(main)
        feats = {"level_0": level0, "level_1": _views(f1, B, V, H // 2, W // 2),
                 "level_2": f0.reshape_views(B, V) if isinstance(f0, convnet.LookupRecords) else f0.reshape(B, V, -1, H, W)}
        return feats, st0

    # ------------------------------------------------------------------ forward
    @staticmethod
    def ensure_rays(batch):
        """batch['rays_i'] for every rendered level that the batch does not carry: built on the device from the target
        camera (ops.make_rays = the full-image branch of lib/datasets/enerf_utils.py:25-71), which removes the
        10.5 MB host->device copy per level and frame a loader-built ray tensor costs.  Batches that do carry rays
        (training: sampled rays / patches) are used as they are."""
        cc = cfg.enerf.cas_config
        src = batch["all_src_inps"] if "all_src_inps" in batch and "src_inps" not in batch else batch["src_inps"]
        H, W = src.shape[-2:]
        for i in range(cc.num):
            have = batch.get(f"rays_{i}")
            # (rays this function built on an earlier call are rebuilt: the caller may have moved the camera in place)
            if cc.render_if[i] and (have is None or getattr(have, "_bmv_built_rays", False)):
                batch[f"rays_{i}"] = ops.make_rays(batch["tar_ext"], batch["tar_ixt"], H, W, cc.render_scale[i])
        return batch

    def forward(self, batch):
        """Inference on the GPU replays a HIP graph of the frame from the second call with the same shapes on
        (autograph.AutoGraph: the drop-in call itself, not a separate harness; BMV_AUTOGRAPH=0 keeps every call eager);
        training and everything else run the launches one by one."""
        if self._autograph.usable(batch):
            return self._autograph(batch)
        return self._forward_checked(batch)

    def _forward_checked(self, batch):
        try:
            return self._forward(batch)
        finally:
            self.set_volume_records(False)       # the modules go back to planar outputs for any other caller
            if self._f0_pending is not None:     # (a level that is not rendered, an exception: join before leaving)
                if torch.cuda.is_available():
                    torch.cuda.current_stream().wait_stream(self._f0_pending)
                self._f0_pending = None

    def _autograph_inputs(self, batch):
        """The batch tensors an inference frame of THIS network reads (autograph copies only these into its captured
        inputs; `all_src_*`, the unrendered levels' rays, targets and masks ride along in a loader's batch unread)."""
        cc = cfg.enerf.cas_config
        return {"src_inps", "src_exts", "src_ixts", "tar_ext", "tar_ixt", "near_far"} | {
            f"rays_{i}" for i in range(cc.num) if cc.render_if[i]}

    def _autograph_deferrable(self, batch):
        """The large inputs whose consumers can read them through a pointer table when the frame is replayed
        (ops.PtrTable): the images (conv0_fused, the lookup records' colours in fpn_smooth) and the rendered levels'
        rays (the fused renderer).  A statement of intent -- autograph verifies it on the captured frame."""
        cc = cfg.enerf.cas_config
        return ("src_inps",) + tuple(f"rays_{i}" for i in range(cc.num) if cc.render_if[i])

    def _autograph_key(self, batch):
        """What a captured frame is specialised to besides shapes and parameters: the execution switches of this module
        (tests and tuning scripts flip them between calls)."""
        return (self.sweep_algo, self.overlap_front, self.lookup_records, self.frame_setup, self.volume_records,
                self.overlap_eager, self.cost_reg_0.split_bf16, self.cost_reg_1.split_bf16,
                self.cost_reg_0.conv_c4, self.cost_reg_1.conv_c4, self.cost_reg_0.quad_volume, self.cost_reg_1.quad_volume,
                self.cost_reg_0.conv_c4s, self.cost_reg_1.conv_c4s, self.defer_f0, switches.get("BMV_FPN_S"), switches.get("BMV_CONV0_S"), switches.get("BMV_CONV2D_S"), switches.get("BMV_CONV2D_S_REC"))

    def _apply(self, fn, *args, **kwargs):       # .to() / .cuda() / .float() replace storage: captured graphs are stale
        ag = self.__dict__.get("_autograph")
        if ag is not None:
            ag.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def _forward(self, batch):
        cc = cfg.enerf.cas_config
        self.ensure_rays(batch)
        views = (batch["src_inps"], batch["src_exts"], batch["src_ixts"])
        st0 = None
        if self._side_stream is None and batch["src_inps"].is_cuda:
            self._side_stream = torch.cuda.Stream(priority=switches.get("BMV_SIDE_PRIO"))   # created outside any capture
        self.feature_net.pack_lookup = (self.wants_lookup_records() and engine_ok(self.feature_net, batch["src_inps"])
                                        and 2 <= batch["src_inps"].shape[1] <= 4)     # (the fused renderer's view counts)
        self.set_volume_records(self.feature_net.pack_lookup)
        try:
            if (self.overlap_front and batch["src_inps"].is_cuda
                    and (self.overlap_eager or torch.cuda.is_current_stream_capturing())
                    and engine_ok(self.feature_net, batch["src_inps"])):
                feats, st0 = self._front_overlapped(batch, views)
            else:
                self._pre = self.camera_only(views, batch) if (self.frame_setup and not self.wants_grad()
                                                               and batch["src_inps"].is_cuda) else None
                feats = self.forward_feat(batch["src_inps"])
        finally:
            self.feature_net.pack_lookup = False
        render = self.render_level_train if self.wants_grad() else self.render_level
        ret = {}
        st = None
        for i in range(cc.num):
            st = st0 if (i == 0 and st0 is not None) else self.level_front(i, feats[f"level_{i}"], views, batch, st,
                                                                           pre=self._pre)
            if not cc.render_if[i]:
                continue
            if self._f0_pending is not None:        # (the deferred full-resolution map: the renderer is its first reader)
                torch.cuda.current_stream().wait_stream(self._f0_pending)
                self._f0_pending = None
            rgb, depth, weights = render(i, st, feats[f"level_{cc.render_im_feat_level[i]}"], views, batch)
            ret_i = {"rgb": rgb, "depth": depth, "weights": weights,
                     "depth_mvs": torch.reciprocal(st.depth) if cc.depth_inv[i] else st.depth, "std": st.std}
            if switches.on("BMV_CHECK_NAN") and bool(rgb.isnan().any()):
                raise FloatingPointError(f"NaN in rgb_level{i}")   # reference: ipdb trap, network.py:110-111
            ret.update({f"{k}_level{i}": v for k, v in ret_i.items()})
        return ret
