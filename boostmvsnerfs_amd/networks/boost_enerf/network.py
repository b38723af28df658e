"""BoostMVSNeRFs on the ENeRF backbone: K cost volumes (view triplets chosen by
a 3-D visibility criterion) are rendered per sample and fused by
visibility-weighted alpha compositing.

Boundary of lib/networks/boost_enerf/network.py:10-237: `Network(preprocess)`,
`forward(batch)` (needs `<cfg.result_dir>/view_selection.json`),
`forward_view_selection(batch)` -> {f"{scene}_{tar_view}": [triplet ids]}.

Kernel plan per rendered level: for each of the K volumes one fused render launch
in MLP-only mode (raw [rgb, sigma], sample depths and the viewport visibility
fraction come out of the same kernel), then ONE blend kernel that normalises the
masks over K, fuses alphas/colours and composites -- the (B,K,N,Ns,*) stacks are
written once and read once.  View selection needs sample positions only, so the
MLP evaluation the reference performs and discards (network.py:53-59) is skipped.
"""
import itertools
import json
import os

import torch

from ... import convnet, ops, switches
from ...config import cfg
from ..enerf import network as enerf_network


def view_triplets(n_views, per_volume=3):
    """torch.combinations(arange(N), 3) order (lexicographic)."""
    return list(itertools.combinations(range(n_views), per_volume))


def greedy_cover(masks, k):
    """search_k_best_views (network.py:71-95): masks (T,H,W) on any device -> list of <= k ids.
    Greedy set cover on the visibility maps: repeatedly take the triplet that adds most
    not-yet-covered visibility; stop early when nothing adds any."""
    T, H, W = masks.shape
    remaining = torch.ones_like(masks[0])
    chosen = []
    for _ in range(k):
        gains = ((masks * remaining).sum((1, 2)) / (H * W)).tolist()
        best, best_id = 0.0, None
        for i, g in enumerate(gains):
            if i in chosen:
                continue
            if g > best:
                best, best_id = g, i
        if best_id is None:
            break
        remaining = remaining * (1 - masks[best_id])
        chosen.append(best_id)
    return chosen or [0]


class Network(enerf_network.Network):
    def __init__(self, preprocess=False):
        super().__init__()
        self.view_selection_outputs = None
        self.capture = None
        self._sel_cache = {}
        self._streams = []
        self._cam_pre = None
        self.side_setup = switches.on("BMV_BOOST_SIDE_SETUP")
        self.parallel_volumes = switches.on("BMV_BOOST_STREAMS")
        # the K cost volumes as one batch through the regularisers instead of K chains on K streams (round 3; opt-in:
        # measured 3.29 ms against 3.15 ms per 480x736 K = 4 frame -- the frame is 4 x 0.7 ms of render launches, and
        # under K streams the regularisers' short launches already hide under the other volumes' renders)
        self.batched_volumes = switches.on("BMV_BOOST_BATCHED")
        # multi-GPU, `--shard volumes` (boostmvsnerfs_amd/sharding.py VolumeShard): build and render only these cost
        # volumes (indices into the K selected ones) and return their stacked (raw, z, mask) instead of the fused picture
        self.volume_ids = None
        self.by_index = True        # inference: views picked by index inside the kernels (tests compare with gathered copies)
        if not preprocess:
            path = os.path.join(cfg.result_dir, "view_selection.json")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: run the view-selection preprocess first "
                                        "(forward_view_selection, run.py:39-85)")
            with open(path, "r") as f:
                self.view_selection_outputs = json.load(f)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _pick(batch, ids):
        """ids: (B,S) long tensor of view indices -> the S views of every batch item."""
        bi = torch.arange(ids.shape[0], device=ids.device)[:, None]
        return (batch["all_src_inps"][bi, ids], batch["all_src_exts"][bi, ids], batch["all_src_ixts"][bi, ids])

    @staticmethod
    def _pick_host(batch, ids):
        """_pick for ONE batch item with the view numbers known on the host (the triplets of view_selection.json): three
        concatenations instead of an arange + three index kernels + their index arithmetic (8 launches, 19 us for the
        12.7 MB of images alone at 6 x 480 x 736: the tail of every K-volume frame)."""
        return tuple(torch.stack([batch[k][:, i] for i in ids], 1) for k in ("all_src_inps", "all_src_exts", "all_src_ixts"))

    @staticmethod
    def _pick_feats(f, bi, ids):
        """f[bi, ids] that keeps a channel-last feature buffer channel-last (the sweep reads it without a transpose)."""
        if isinstance(f, ops.QuadFeats):           # quad-planar (the inference sweep's layout): gathered as it is
            return ops.QuadFeats(f.data[bi, ids].contiguous())
        cl = f.permute(0, 1, 3, 4, 2)
        if not f.is_contiguous() and cl.is_contiguous():
            return cl[bi, ids].permute(0, 1, 4, 2, 3)
        return f[bi, ids]

    # ------------------------------------------------------------------ view selection (a17)
    def calc_mask(self, src_views_id, batch, feats=None):
        """2-D visibility map of one triplet for every rendered level (network.py:22-69)."""
        cc = cfg.enerf.cas_config
        B = batch["all_src_inps"].shape[0]
        ids = torch.as_tensor(src_views_id, device=batch["all_src_inps"].device).view(1, -1).expand(B, -1)
        views = self._pick(batch, ids)
        if feats is None:
            feats = self.forward_feat(views[0])
            pick = lambda f: f
        else:
            bi = torch.arange(B, device=ids.device)[:, None]
            pick = lambda f: self._pick_feats(f, bi, ids)
        H, W = views[0].shape[-2:]
        out, st = {}, None
        for i in range(cc.num):
            st = self.level_front(i, pick(feats[f"level_{i}"]), views, batch, st)
            if not cc.render_if[i]:
                continue
            rs = cc.render_scale[i]
            Hr, Wr = int(H * rs), int(W * rs)
            Ns = cc.num_samples[i]
            rays = ops.build_rays(batch[f"rays_{i}"], st.depth, st.std, st.near_far, Hr, Wr, cc.depth_inv[i])
            xyz, _, z = ops.sample_along_depth(rays, Ns, cc.depth_inv[i])
            vis = ops.mask_viewport(xyz, views[1], views[2], Wr - 1, Hr - 1).view(B, -1, Ns) / Ns
            # the reference composites the mask as if it were [rgb, sigma] and keeps the mean colour
            rgb, _, _ = ops.composite(vis[..., None].expand(-1, -1, -1, 4).contiguous(), z, cfg.enerf.white_bkgd)
            out[f"mask_level{i}"] = rgb.mean(-1).view(B, Hr, Wr)
        return out

    def search_k_best_views(self, masks, k, level):
        T = len(masks)
        stack = torch.stack([masks[f"mask_level{level}_view{i}"] for i in range(T)])
        return greedy_cover(stack.reshape(T, *stack.shape[-2:]), k)

    def forward_view_selection(self, batch):
        cc = cfg.enerf.cas_config
        N = batch["all_src_inps"].shape[1]
        with torch.no_grad():
            feats = self.forward_feat(batch["all_src_inps"])      # once for all N views, not per triplet
            all_masks = {}
            for t, ids in enumerate(view_triplets(N, 3)):
                m = self.calc_mask(ids, batch, feats)
                all_masks.update({f"{k}_view{t}": v for k, v in m.items()})
        result = {}
        for i in range(cc.num):
            if not cc.render_if[i]:
                continue
            level_masks = {k: v for k, v in all_masks.items() if k.startswith(f"mask_level{i}")}
            sel = [int(s) for s in self.search_k_best_views(level_masks, cc.k_best, i)]
            for scene, view in zip(batch["meta"]["scene"], batch["meta"]["tar_view"]):
                result[f"{scene}_{view}"] = sel
        return result

    # ------------------------------------------------------------------ fused forward (a15, a16)
    def merge_mlp_outputs(self, raws, masks, z_vals):
        """(B,K,N,Ns,4), (B,K,N,Ns), (B,K,N,Ns) -> fused rgb/depth/weights (network.py:163-170 +
        utils.py:639-667); mask normalisation over K happens inside the kernel."""
        if cfg.enerf.white_bkgd:
            raise NotImplementedError          # as the reference (utils.py:660-661)
        if raws.requires_grad:                  # fine-tuning: masks are constants (built under no_grad)
            from ...autograd import Blend
            with torch.no_grad():
                tot = masks.sum(1, keepdim=True)
                masks = torch.where(tot > 0, masks / tot, torch.full_like(masks, 1.0 / masks.shape[1]))
            rgb, depth, weights = Blend.apply(raws, masks, z_vals)
        else:
            rgb, depth, weights = ops.blend(raws, masks, z_vals, normalise=True)
        return {"rgb": rgb, "depth": depth, "weights": weights}

    def _forward_parallel(self, batch, feats, sel, sel32, cams, K, late=None):
        """The K cost volumes are independent until the fusion: each one's chain (sweeps, regularisers, depth
        regression, render in MLP-only mode) runs on its own HIP stream; the deep U-Net levels are launches of a few
        dozen workgroups that fill the chip only together.  The main stream joins them before the blend."""
        cc = cfg.enerf.cas_config
        dev = batch["all_src_inps"].device
        main = torch.cuda.current_stream()
        while len(self._streams) < K:
            self._streams.append(torch.cuda.Stream())
        ks = list(range(K)) if self.volume_ids is None else [int(k) for k in self.volume_ids]
        stacks = {}
        for i in range(cc.num):
            if cc.render_if[i]:
                n_i, ns_i = batch[f"rays_{i}"].shape[1], cc.num_samples[i]
                stacks[i] = (torch.empty(1, len(ks), n_i, ns_i, 4, device=dev), torch.empty(1, len(ks), n_i, ns_i, device=dev),
                             torch.empty(1, len(ks), n_i, ns_i, device=dev))
        # the packed-weight caches (MLP blobs, folded convolution weights) are filled lazily by whichever chain touches
        # them first: fill them HERE, on the stream every chain forks from, or volumes 1..K-1 could read blobs that
        # volume 0's stream is still writing (first frame after load / .to() / an optimiser step)
        for i in range(cc.num):
            getattr(self, f"cost_reg_{i}").prepack()
            if cc.render_if[i]:
                getattr(self, f"nerf_{i}").packed_weights()
        pre = getattr(self, "_cam_pre", None)      # per volume: projection matrices + level-0 hypotheses (_forward_boost)
        first = {}
        fronts = {}
        if late is not None:
            # FeatureNet's coarsest map is all the level-0 cost volumes need (enerf.Network._front_overlapped, K times): the K
            # level-0 chains -- sweep, 3-D regulariser, depth regression: short launches that fill the chip only together --
            # run on their streams UNDER FeatureNet's top-down path (3 large launches on the main stream, 0.23 of a 2.2 ms
            # frame at 6 x 480 x 736), instead of behind it
            for j, k in enumerate(ks):
                s = self._streams[j]
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    fronts[j] = self.level_front(0, feats["level_0"], (batch["all_src_inps"], *cams[k]), batch, None,
                                                 view_ids=sel32[k], pre=pre[k] if pre is not None else None)
            feats.update(late())
        for j, k in enumerate(ks):
            s = self._streams[j]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                vid = sel32[k]
                views = (batch["all_src_inps"], *cams[k])
                st = fronts.get(j)
                for i in range(cc.num):
                    if not (i == 0 and st is not None):
                        st = self.level_front(i, feats[f"level_{i}"], views, batch, st, view_ids=vid,
                                              pre=pre[k] if pre is not None else None)
                    if cc.render_if[i]:
                        self.render_level(i, st, feats[f"level_{cc.render_im_feat_level[i]}"], views, batch, mode=1,
                                          outs=tuple(t[:, j] for t in stacks[i]), view_ids=vid)
                        if k == 0:
                            first[i] = (st.depth, st.std)
        for j in range(len(ks)):
            main.wait_stream(self._streams[j])
        ret = {}
        for i, (raws, zs, ms) in stacks.items():
            if self.ray_range is not None:   # the render launches wrote rays [begin, end) of the full-size buffers
                b_, e_ = self.ray_range
                raws, zs, ms = raws[:, :, b_:e_], zs[:, :, b_:e_], ms[:, :, b_:e_]
            if self.capture is not None:     # tests: per-volume raw outputs / depths / visibility masks
                self.capture[f"level{i}"] = (raws, zs, ms)
            if self.volume_ids is not None:  # volume sharding: the fusion happens after the exchange between ranks
                out = {"stacks": (raws, zs, ms)}
            else:
                out = self.merge_mlp_outputs(raws.contiguous(), ms.contiguous(), zs.contiguous())
            if i in first:
                depth0, std0 = first[i]                             # depth_mvs / std come from volume 0 only
                if not torch.cuda.is_current_stream_capturing():
                    depth0.record_stream(main), std0.record_stream(main)
                out["depth_mvs"] = torch.reciprocal(depth0) if cc.depth_inv[i] else depth0
                out["std"] = std0
            ret.update({f"{k_}_level{i}": v for k_, v in out.items()})
        # the reference leaves the last triplet in batch['src_*'] (network.py:196-198)
        last = getattr(self, "_last_triplet", None)
        batch["src_inps"], batch["src_exts"], batch["src_ixts"] = (self._pick_host(batch, last) if last is not None
                                                                    else self._pick(batch, sel[:, K - 1]))
        return ret

    def _forward_batched(self, batch, feats, sel, sel32, cams, K):
        """The K cost volumes as ONE batch (round 3): per cascade level K sweeps write the slices of a (K,C,D,h,w)
        variance tensor, then ONE regulariser pass with batch K -- every launch of the U-Net carries K times the
        workgroups of a single volume's, so its latency-bound interior layers (8-10 us each whatever their size) are
        paid once per level instead of K times side by side on K streams -- one depth regression, one hypothesis
        kernel, and K render launches (each fills the chip on its own).  Everything on the current stream."""
        cc = cfg.enerf.cas_config
        dev = batch["all_src_inps"].device
        ks = list(range(K))
        src_exts = torch.cat([cams[k][0] for k in ks], 0)               # (K,S,4,4): the K triplets as batch items
        src_ixts = torch.cat([cams[k][1] for k in ks], 0)
        tar_ext = batch["tar_ext"].expand(K, -1, -1).contiguous()
        tar_ixt = batch["tar_ixt"].expand(K, -1, -1).contiguous()
        near_far = batch["near_far"].expand(K, -1).contiguous()
        H, W = batch["all_src_inps"].shape[-2:]
        ret, st = {}, None
        for i in range(cc.num):
            h, w = int(H * cc.volume_scale[i]), int(W * cc.volume_scale[i])
            D = cc.volume_planes[i]
            cur = enerf_network.LevelState()
            if st is None:
                cur.depth_values, cur.near_far = ops.depth_values_uniform(near_far, D, h, w, cc.depth_inv[i])
            else:
                if not cc.depth_inv[i - 1] or cc.depth_inv[i]:
                    raise NotImplementedError("cascade levels must go disparity -> depth")
                cur.depth_values, cur.near_far = ops.depth_values_cascade(st.depth, st.std, st.near_far, h, w, D)
            proj = ops.proj_mats(src_exts, src_ixts, tar_ext, tar_ixt, cc.im_feat_scale[i], cc.volume_scale[i])
            f_i = feats[f"level_{i}"]
            variance = torch.empty(K, f_i.shape[2], D, h, w, device=dev)
            for k in ks:
                ops.sweep_variance_views(f_i, sel32[k], proj[k:k + 1], cur.depth_values[k:k + 1], out=variance[k:k + 1],
                                         plane_uniform=st is None)
            cur.feature_volume, depth_prob = getattr(self, f"cost_reg_{i}")(variance)
            cur.depth, cur.std = ops.depth_regress(depth_prob, cur.depth_values, cc.depth_inv[i])
            st = cur
            if not cc.render_if[i]:
                continue
            n_i, ns_i = batch[f"rays_{i}"].shape[1], cc.num_samples[i]
            stacks = (torch.empty(1, K, n_i, ns_i, 4, device=dev), torch.empty(1, K, n_i, ns_i, device=dev),
                      torch.empty(1, K, n_i, ns_i, device=dev))
            im_feat = feats[f"level_{cc.render_im_feat_level[i]}"]
            fv = st.feature_volume
            for k in ks:
                one = enerf_network.LevelState()
                one.depth, one.std, one.near_far = st.depth[k:k + 1], st.std[k:k + 1], st.near_far[k:k + 1]
                one.feature_volume = type(fv)(fv.t[k:k + 1]) if hasattr(fv, "t") else fv[k:k + 1]
                views = (batch["all_src_inps"], *cams[k])
                self.render_level(i, one, im_feat, views, batch, mode=1, outs=tuple(t[:, k] for t in stacks), view_ids=sel32[k])
            raws, zs, ms = stacks
            if self.ray_range is not None:   # the render launches wrote rays [begin, end) of the full-size buffers
                b_, e_ = self.ray_range
                raws, zs, ms = raws[:, :, b_:e_], zs[:, :, b_:e_], ms[:, :, b_:e_]
            if self.capture is not None:     # tests: per-volume raw outputs / depths / visibility masks
                self.capture[f"level{i}"] = (raws, zs, ms)
            out = self.merge_mlp_outputs(raws.contiguous(), ms.contiguous(), zs.contiguous())
            depth0, std0 = st.depth[:1], st.std[:1]                     # depth_mvs / std come from volume 0 only
            out["depth_mvs"] = torch.reciprocal(depth0) if cc.depth_inv[i] else depth0
            out["std"] = std0
            ret.update({f"{k_}_level{i}": v for k_, v in out.items()})
        batch["src_inps"], batch["src_exts"], batch["src_ixts"] = self._pick(batch, sel[:, K - 1])
        return ret

    def _forward_checked(self, batch):           # (forward itself, with the self-capturing replay, is the base class's)
        try:
            return self._forward_boost(batch)
        finally:
            self.set_volume_records(False)

    def _autograph_inputs(self, batch):
        cc = cfg.enerf.cas_config
        return {"all_src_inps", "all_src_exts", "all_src_ixts", "tar_ext", "tar_ixt", "near_far"} | {
            f"rays_{i}" for i in range(cc.num) if cc.render_if[i]}

    def _autograph_deferrable(self, batch):
        cc = cfg.enerf.cas_config
        return ("all_src_inps",) + tuple(f"rays_{i}" for i in range(cc.num) if cc.render_if[i])

    def _autograph_key(self, batch):
        """A captured K-volume frame is specialised to the cost-volume triplets view_selection.json selects for the
        batch's targets (they are baked into the graph as device constants) and to the capture hook of the tests."""
        base = super()._autograph_key(batch) + (self.parallel_volumes, self.batched_volumes, self.by_index, self.capture is not None,
                                                 switches.get("BMV_BOOST_OVERLAP"))
        if self.view_selection_outputs is None:
            return base
        meta = batch["meta"]
        return base + (tuple(tuple(self.view_selection_outputs[f"{s}_{v}"]) for s, v in zip(meta["scene"], meta["tar_view"])),
                       int(cfg.enerf.cas_config.k_best))

    def _forward_boost(self, batch):
        if self.view_selection_outputs is None:
            raise RuntimeError("Network(preprocess=True) only supports forward_view_selection()")
        cc = cfg.enerf.cas_config
        self.ensure_rays(batch)
        dev = batch["all_src_inps"].device
        B, N = batch["all_src_inps"].shape[:2]
        K = int(cc.k_best)
        # (B,K,3) view ids of the K cost volumes; cached on the device per (targets, N, K): no host->device copy
        # on the frame path, and the forward stays capturable as a HIP graph
        key = (tuple(f"{s}_{v}" for s, v in zip(batch["meta"]["scene"], batch["meta"]["tar_view"])), N, K, str(dev))
        sel = self._sel_cache.get(key)
        if sel is None:
            trip = torch.tensor(view_triplets(N, cfg.enerf.cost_volume_input_views), device=dev)
            picks = [self.view_selection_outputs[t] for t in key[0]]
            n_trip = trip.shape[0]
            if any(not (0 <= int(v) < n_trip) for row in picks for v in row):      # host check: the kernels index with them
                raise ValueError(f"view_selection.json holds a triplet index outside [0, {n_trip}) for {N} source views")
            k_best = torch.tensor(picks, device=dev)
            if k_best.shape[1] < K:
                raise ValueError(f"view_selection.json holds {k_best.shape[1]} volumes per target, cfg k_best={K}")
            sel = trip[k_best[:, :K]]                               # (B,K,3)
            if len(self._sel_cache) > 64:
                self._sel_cache.clear()
            # (+ the last volume's view numbers as Python ints, for the batch['src_*'] the reference leaves behind)
            self._sel_cache[key] = sel = (sel, [int(v) for v in view_triplets(N, cfg.enerf.cost_volume_input_views)[int(picks[0][K - 1])]]
                                          if B == 1 else None)
        sel, self._last_triplet = sel
        # what the K chains need from the cameras alone -- the K triplets' camera matrices, K x S projection matrices per
        # cascade level and level 0's hypotheses -- on a side stream UNDER FeatureNet: one gather per camera tensor and
        # ONE launch (ops.frame_setup, the triplets as batch items) instead of 2 K + K latency-bound launches and ~20 index
        # kernels inside the chains.  (The same launch in front of the fork on the main stream was measured slower in round 3:
        # its single-thread fp64 inversions are pure latency there.)
        cam = None
        if (B == 1 and self.frame_setup and self.side_setup and self.by_index and dev.type == "cuda" and not self.wants_grad()
                and (self.parallel_volumes or self.volume_ids is not None) and not self.batched_volumes):
            main = torch.cuda.current_stream()
            # (on the first volume's own stream, whose chain simply continues behind it.  A separate stream for this
            # one launch crashed hipGraphLaunch on ROCm 7.2 when the process had captured other graphs before --
            # tests/test_gpu_framegraph.py in file order)
            while len(self._streams) < K:
                self._streams.append(torch.cuda.Stream())
            su = self._streams[0]
            su.wait_stream(main)
            with torch.cuda.stream(su):
                H, W = batch["all_src_inps"].shape[-2:]
                ext_all, ixt_all = batch["all_src_exts"][0][sel[0]], batch["all_src_ixts"][0][sel[0]]    # (K,S,4,4), (K,S,3,3)
                proj, (dv0, nf0) = ops.frame_setup(
                    ext_all, ixt_all, batch["tar_ext"].expand(K, -1, -1), batch["tar_ixt"].expand(K, -1, -1),
                    [cc.im_feat_scale[i] for i in range(cc.num)], [cc.volume_scale[i] for i in range(cc.num)],
                    batch["near_far"].expand(K, -1), cc.volume_planes[0], int(H * cc.volume_scale[0]),
                    int(W * cc.volume_scale[0]), cc.depth_inv[0])
            cam = {"ext": ext_all, "ixt": ixt_all, "stream": su,
                   "pre": [{"proj": [p[k:k + 1] for p in proj], "dv0": (dv0[k:k + 1], nf0[k:k + 1])} for k in range(K)],
                   "keep": (*proj, dv0, nf0)}
        # all N views once; inference: the full-resolution map as the fused renderer's lookup records
        self.feature_net.pack_lookup = (self.wants_lookup_records()
                                        and enerf_network.engine_ok(self.feature_net, batch["all_src_inps"]))
        self.set_volume_records(self.feature_net.pack_lookup)
        late = None
        x_all = batch["all_src_inps"]
        fn = self.feature_net
        if (switches.on("BMV_BOOST_OVERLAP") and B == 1 and dev.type == "cuda" and not self.wants_grad() and self.by_index
                and (self.parallel_volumes or self.volume_ids is not None) and not self.batched_volumes
                and cc.num == 2 and not cc.render_if[0] and cc.render_if[1] and cc.render_scale[1] == 1.0
                and enerf_network.engine_ok(fn, x_all)):
            # two phases: the encoder + top layer now, the top-down path (`late`) once the K level-0 chains are on their streams
            _, V, C_, H_, W_ = x_all.shape
            fn.quad_out = self.sweep_algo == 0 or self.sweep_algo >= 500
            xf = x_all.reshape(V, C_, H_, W_)
            pack = fn.pack_lookup
            try:
                c0, c1, p2, p2_cl = fn.engine_bottom_up(xf)
            finally:
                fn.pack_lookup = False
            feats = {"level_0": enerf_network._views(p2_cl, 1, V, H_ // 4, W_ // 4)}

            def late():
                f1, f0 = fn.engine_top_down(c0, c1, p2, rgb=xf if pack else None)
                return {"level_1": enerf_network._views(f1, 1, V, H_ // 2, W_ // 2),
                        "level_2": f0.reshape_views(1, V) if isinstance(f0, convnet.LookupRecords) else f0.reshape(1, V, -1, H_, W_)}
        else:
            try:
                feats = self.forward_feat(x_all)
            finally:
                self.feature_net.pack_lookup = False
        if cam is not None:
            # (long done: FeatureNet took 100x its time.  The tensors allocated on the side stream are read on the main
            # and the volume streams and released when this function returns; the side stream's next allocation comes
            # after its wait for the main stream of the NEXT frame, which has joined every volume stream by then: no
            # record_stream needed)
            torch.cuda.current_stream().wait_stream(cam["stream"])
        bi = torch.arange(B, device=dev)[:, None]
        states = [None] * K
        ret = {}
        train = self.wants_grad()
        # Inference with the engine's channel-last feature maps: the sweep and render kernels pick each volume's three
        # views out of the all-views tensors by index -- no gathered copies of images / feature maps per volume
        # (those copies were 15 % of a K = 4 frame); only the 4x4 / 3x3 camera matrices are gathered, once per volume.
        by_index = (self.by_index and not train and all(cc.render_scale[i] == 1.0 for i in range(cc.num) if cc.render_if[i])
                    and all(isinstance(feats[f"level_{i}"], ops.QuadFeats)
                            or (not feats[f"level_{i}"].is_contiguous()
                                and feats[f"level_{i}"].permute(0, 1, 3, 4, 2).is_contiguous())
                            for i in range(cc.num) if f"level_{i}" in feats))   # (two-phase FeatureNet: the engine's layouts)
        assert late is None or by_index
        if by_index:
            # (B,K,3) -> K tensors (B,3); in range by construction: rows of combinations(range(N), 3) picked by the
            # triplet numbers validated above, so no device read is spent on ops.check_view_ids
            sel32 = [ops.mark_view_ids(sel[:, k].to(torch.int32).contiguous(), N) for k in range(K)]
            if cam is not None:
                cams = [(cam["ext"][k:k + 1], cam["ixt"][k:k + 1]) for k in range(K)]
                self._cam_pre = cam["pre"]
            elif B == 1:   # one gather for the K triplets' cameras: (K,S,4,4) / (K,S,3,3), volume k = a view of row k
                ext_all, ixt_all = batch["all_src_exts"][0][sel[0]], batch["all_src_ixts"][0][sel[0]]
                cams = [(ext_all[k:k + 1], ixt_all[k:k + 1]) for k in range(K)]
                self._cam_pre = None
            else:
                self._cam_pre = None
                cams = [(batch["all_src_exts"][bi, sel[:, k]], batch["all_src_ixts"][bi, sel[:, k]]) for k in range(K)]
            if self.batched_volumes and self.volume_ids is None and B == 1:
                return self._forward_batched(batch, feats, sel, sel32, cams, K)
            if (self.parallel_volumes or self.volume_ids is not None) and B == 1:
                try:
                    return self._forward_parallel(batch, feats, sel, sel32, cams, K, late=late)
                finally:
                    self._cam_pre = None      # (tensors of this frame / this capture's pool: not kept across calls)
        if self.volume_ids is not None:
            raise NotImplementedError("volume_ids (multi-GPU volume sharding) needs the inference path with views by index, B = 1")
        for i in range(cc.num):
            raws, zs, ms = [], [], []
            stacks = None
            if cc.render_if[i] and B == 1 and not train:   # K render launches write straight into the stacked buffers
                n_i, ns_i = batch[f"rays_{i}"].shape[1], cc.num_samples[i]
                stacks = (torch.empty(1, K, n_i, ns_i, 4, device=dev), torch.empty(1, K, n_i, ns_i, device=dev),
                          torch.empty(1, K, n_i, ns_i, device=dev))
            for k in range(K):
                ids = sel[:, k]
                if by_index:
                    vid = sel32[k]
                    views = (batch["all_src_inps"], *cams[k])
                    states[k] = self.level_front(i, feats[f"level_{i}"], views, batch, states[k], view_ids=vid)
                    if not cc.render_if[i]:
                        continue
                    im_feat = feats[f"level_{cc.render_im_feat_level[i]}"]
                    o = tuple(t[:, k] for t in stacks) if (stacks is not None and self.ray_range is None) else None
                    r = self.render_level(i, states[k], im_feat, views, batch, mode=1, outs=o, view_ids=vid)
                    if o is None:
                        raws.append(r[0]), zs.append(r[1]), ms.append(r[2])
                    continue
                views = self._pick(batch, ids)
                states[k] = self.level_front(i, self._pick_feats(feats[f"level_{i}"], bi, ids), views, batch, states[k])
                if not cc.render_if[i]:
                    continue
                im_feat = feats[f"level_{cc.render_im_feat_level[i]}"][bi, ids]
                if train:
                    raw, z, m = self.render_level_train(i, states[k], im_feat, views, batch, mode=1)
                    raws.append(raw), zs.append(z), ms.append(m)
                elif stacks is not None and self.ray_range is None:
                    self.render_level(i, states[k], im_feat, views, batch, mode=1, outs=tuple(t[:, k] for t in stacks))
                else:
                    raw, z, m = self.render_level(i, states[k], im_feat, views, batch, mode=1)
                    raws.append(raw), zs.append(z), ms.append(m)
            if not cc.render_if[i]:
                continue
            if raws:
                stacks = (torch.stack(raws, 1), torch.stack(zs, 1), torch.stack(ms, 1))
            if self.capture is not None:     # tests: per-volume raw outputs / depths / visibility masks
                self.capture[f"level{i}"] = stacks
            out = self.merge_mlp_outputs(stacks[0], stacks[2], stacks[1])
            st0 = states[0]                                         # depth_mvs / std come from volume 0 only
            out["depth_mvs"] = torch.reciprocal(st0.depth) if cc.depth_inv[i] else st0.depth
            out["std"] = st0.std
            ret.update({f"{k_}_level{i}": v for k_, v in out.items()})
        # the reference leaves the last triplet in batch['src_*'] (network.py:196-198)
        batch["src_inps"], batch["src_exts"], batch["src_ixts"] = self._pick(batch, sel[:, K - 1])
        return ret
