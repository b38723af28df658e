"""CPU oracle for the MVSNeRF / BoostMVSNeRFs(MVSNeRF) rendering hot path.

TEST INFRASTRUCTURE ONLY (same rules as oracle/enerf.py): a from-scratch fp32
torch-CPU restatement of the reference, imported only by tests/, smoke() and
bench.py's cpu_baseline leg.

Parity pin: checked function by function against outputs of the reference
itself through tests/golden/mvsnerf_tiny.npz / boost_mvsnerf_tiny.npz
(tests/golden/make_golden.py, tests/test_oracle_golden_mvs.py).  One reference
behaviour cannot be pinned as is: build_volume_costvar_img fills channels 0-2 of
its padded volume from `torch.empty` (mvsnerf/network.py:912-914), so the padding
border holds whatever the allocator returns.  The golden generator zero-fills
that allocation; this oracle and the HIP kernel define the border as zeros.

Citations are into /root/reference/lib/networks/mvsnerf/ unless noted.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import enerf as E

PAD = 24  # network.py:1106, 1016


# ---------------------------------------------------------------------------
# CNNs (not hot-path kernels): InPlaceABN = batch norm + leaky_relu(0.01)
#   FeatureNet network.py:699-733, CostRegNet network.py:735-779
# ---------------------------------------------------------------------------

def _abn(sd, name, x):
    y = F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                     sd[name + ".bias"], False, 0.0, 1e-5)
    return F.leaky_relu(y, 0.01)


def feature_net(sd, imgs, p="feature."):
    B, V, C, H, W = imgs.shape
    x = imgs.reshape(B * V, C, H, W)
    plan = (("conv0.0", 1, 1), ("conv0.1", 1, 1), ("conv1.0", 2, 2), ("conv1.1", 1, 1), ("conv1.2", 1, 1),
            ("conv2.0", 2, 2), ("conv2.1", 1, 1), ("conv2.2", 1, 1))
    for name, stride, pad in plan:
        x = _abn(sd, p + name + ".bn", F.conv2d(x, sd[p + name + ".conv.weight"], None, stride, pad))
    x = F.conv2d(x, sd[p + "toplayer.weight"], sd[p + "toplayer.bias"])
    return x.view(B, V, 32, H // 4, W // 4)


def cost_reg(sd, x, p="cost_reg_2."):
    def cbr(name, t, stride=1):
        return _abn(sd, p + name + ".bn", F.conv3d(t, sd[p + name + ".conv.weight"], None, stride, 1))

    def up(name, t):
        y = F.conv_transpose3d(t, sd[p + name + ".0.weight"], None, stride=2, padding=1, output_padding=1)
        return _abn(sd, p + name + ".1", y)

    c0 = cbr("conv0", x)
    c2 = cbr("conv2", cbr("conv1", c0, 2))
    c4 = cbr("conv4", cbr("conv3", c2, 2))
    y = cbr("conv6", cbr("conv5", c4, 2))
    y = c4 + up("conv7", y)
    y = c2 + up("conv9", y)
    return c0 + up("conv11", y)


# ---------------------------------------------------------------------------
# a18 get_proj_mats                                         network.py:1070-1090
# ---------------------------------------------------------------------------

def proj_mats(src_exts, src_ixts):
    """View 0 is the reference view: P_i = (K_i/4 E_i) inverse(K_0/4 E_0), P_0 = I -> (B,S,3,4)."""
    B, S = src_exts.shape[:2]
    out = torch.zeros(B, S, 3, 4)
    for b in range(B):
        ref_inv = None
        for i in range(S):
            K = src_ixts[b, i].clone()
            K[:2] = K[:2] * 0.25
            P = torch.eye(4)
            P[:3, :4] = K @ src_exts[b, i, :3, :4]
            if i == 0:
                ref_inv = torch.inverse(P)
                out[b, i] = torch.eye(4)[:3]
            else:
                out[b, i] = (P @ ref_inv)[:3]
    return out


def depth_planes(depth_ranges, D):
    """network.py:1100-1104: D planes linear in depth between 0.8*min and 1.2*max of the ranges."""
    near, far = depth_ranges.min() * 0.8, depth_ranges.max() * 1.2
    t = torch.linspace(0.0, 1.0, D)
    return near * (1.0 - t) + far * t, near, far


# ---------------------------------------------------------------------------
# a19 homo_warp (padded, no z clamp)                        utils.py:580-630
# ---------------------------------------------------------------------------

def warp_grid(proj, depth_values, h, w, pad):
    """proj (B,3,4), depth_values (B,D) -> grid (B,D,hp,wp,2) normalised to the UNPADDED (h,w) source."""
    B, D = depth_values.shape
    hp, wp = h + 2 * pad, w + 2 * pad
    ys, xs = torch.meshgrid(torch.arange(hp, dtype=torch.float32) - pad, torch.arange(wp, dtype=torch.float32) - pad,
                            indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], 0).reshape(1, 3, hp * wp).expand(B, -1, -1)
    rot = (proj[:, :, :3] @ pix).repeat(1, 1, D)
    dv = depth_values[:, :, None, None].repeat(1, 1, hp, wp).reshape(B, 1, D * hp * wp)
    p = rot + proj[:, :, 3:] / dv
    xy = p[:, :2] / p[:, 2:]
    gx = xy[:, 0] / ((w - 1) / 2) - 1
    gy = xy[:, 1] / ((h - 1) / 2) - 1
    return torch.stack([gx, gy], -1).view(B, D, hp, wp, 2)


def warp(src, grid):
    B, D, hp, wp, _ = grid.shape
    out = F.grid_sample(src, grid.view(B, D, hp * wp, 2), mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(B, -1, D, hp, wp)


# ---------------------------------------------------------------------------
# a20 build_volume_costvar_img                              network.py:887-942
# ---------------------------------------------------------------------------

def cost_volume(imgs, feats, projs, depth_values, pad=PAD):
    """imgs (B,3,3,H,W) in [-1,1], feats (B,3,32,h,w), projs (B,3,3,4), depth_values (B,D)
    -> (B, 9+32, D, h+2pad, w+2pad): ref rgb | 2 warped src rgb | masked variance of the features."""
    B, V, C, h, w = feats.shape
    D = depth_values.shape[1]
    hp, wp = h + 2 * pad, w + 2 * pad
    small = F.interpolate(imgs.reshape(B * V, *imgs.shape[2:]), (h, w), mode="bilinear", align_corners=False)
    small = small.view(B, V, 3, h, w)
    out = torch.zeros(B, 9 + C, D, hp, wp)                      # border of ch 0-2: zeros by definition (see header)
    out[:, :3, :, pad:pad + h, pad:pad + w] = small[:, 0, :, None].expand(-1, -1, D, -1, -1)
    ref = F.pad(feats[:, 0], (pad, pad, pad, pad))[:, :, None].repeat(1, 1, D, 1, 1)
    acc, acc2 = ref, ref ** 2
    count = torch.ones(B, 1, D, hp, wp)
    for i in range(1, V):
        grid = warp_grid(projs[:, i], depth_values, h, w, pad)
        wf = warp(feats[:, i], grid)
        out[:, 3 * i:3 * i + 3] = warp(small[:, i], grid)
        inside = ((grid > -1.0) & (grid < 1.0)).all(-1)
        count = count + inside[:, None].float()
        acc = acc + wf
        acc2 = acc2 + wf ** 2
    inv = 1.0 / count
    out[:, 9:] = acc2 * inv - (acc * inv) ** 2
    return out


# ---------------------------------------------------------------------------
# a21 ray_marcher                                           network.py:945-958
# ---------------------------------------------------------------------------

def ray_march(rays, Ns):
    """near/far are rays[..., 6] / rays[..., 7] verbatim (with the shipped loaders: pixel x, y)."""
    near, far = rays[..., 6:7], rays[..., 7:8]
    t = torch.linspace(0.0, 1.0, Ns)
    z = near * (1.0 - t) + far * t
    xyz = rays[..., None, :3] + rays[..., None, 3:6] * z[..., None]
    return xyz, z


# ---------------------------------------------------------------------------
# a22 get_ndc_coordinate                                    utils.py:112-146
# ---------------------------------------------------------------------------

def ndc_coordinate(w2c, K, pts, inv_scale, near, far, pad):
    """pts (N,Ns,3) -> (N,Ns,3): (u, v) in [0,1] of the reference view (re-mapped into the padded
    volume when pad > 0) and depth normalised by [near, far]."""
    N, Ns = pts.shape[:2]
    p = pts.reshape(-1, 3) @ w2c[:3, :3].t() + w2c[:3, 3].reshape(1, 3)
    q = p @ K.t()
    uv = q[:, :2] / q[:, 2:] / inv_scale.reshape(1, 2)
    d = (q[:, 2] - near) / (far - near)
    u, v = uv[:, 0], uv[:, 1]
    if pad > 0:
        wf, hf = (inv_scale + 1) / 4.0
        v = v * hf / (hf + pad * 2) + pad / (hf + pad * 2)
        u = u * wf / (wf + pad * 2) + pad / (wf + pad * 2)
    return torch.stack([u, v, d], -1).view(N, Ns, 3)


# ---------------------------------------------------------------------------
# a23 gen_dir_feature / gen_pts_feats / index_point_feature / build_color_volume
#     renderer.py:111-137, utils.py:300-332, 357-383
# ---------------------------------------------------------------------------

def dir_feature(w2c_ref, rays_d):
    n = rays_d.norm(dim=-1, keepdim=True)
    return (rays_d / n) @ w2c_ref[:3, :3].t()


def volume_lookup(volume, ndc):
    """volume (1,8,D,hp,wp), ndc (N,Ns,3) in [0,1] -> (N,Ns,8); trilinear, zeros padding."""
    g = ndc[None, None] * 2.0 - 1.0
    return F.grid_sample(volume, g, align_corners=True, mode="bilinear")[0, :, 0].permute(1, 2, 0)


def colour_lookup(pts, w2cs, Ks, imgs):
    """pts (N,Ns,3); per view: rgb (bilinear, border) + inside flag -> (N,Ns,4*V)."""
    V, _, H, W = imgs.shape[1:]
    inv_scale = torch.tensor([W - 1.0, H - 1.0])
    out = []
    for i in range(V):
        uvd = ndc_coordinate(w2cs[i], Ks[i], pts, inv_scale, 2.0, 6.0, 0)
        g = uvd[None, ..., :2] * 2.0 - 1.0
        rgb = F.grid_sample(imgs[:, i], g, align_corners=True, mode="bilinear", padding_mode="border")[0].permute(1, 2, 0)
        inside = ((g > -1.0) & (g < 1.0)).all(-1)[0].float()
        out += [rgb, inside[..., None]]
    return torch.cat(out, -1)


# ---------------------------------------------------------------------------
# a24 Embedder.embed                                        network.py:54-58
# ---------------------------------------------------------------------------

def embed(x, n_freq=10):
    freqs = 2.0 ** torch.linspace(0.0, n_freq - 1, n_freq)
    scaled = (x[..., None, :] * freqs[:, None]).reshape(*x.shape[:-1], -1)
    return torch.cat([x, torch.sin(scaled), torch.cos(scaled)], -1)


# ---------------------------------------------------------------------------
# a25 Renderer_ours.forward                                 network.py:201-229
# ---------------------------------------------------------------------------

def renderer_mlp(sd, x, p="nerf.nerf."):
    """x (..., 63+20+3) -> (..., 4) = [sigmoid rgb, relu alpha]."""
    pts, feat, views = x[..., :63], x[..., 63:83], x[..., 83:86]

    def lin(name, t):
        return F.linear(t, sd[p + name + ".weight"], sd[p + name + ".bias"])

    bias = lin("pts_bias", feat)
    h = pts
    for i in range(6):
        h = F.relu(lin(f"pts_linears.{i}", h) * bias)
        if i == 4:
            h = torch.cat([pts, h], -1)
    alpha = F.relu(lin("alpha_linear", h))
    h = F.relu(lin("views_linears.0", torch.cat([lin("feature_linear", h), views], -1)))
    return torch.cat([torch.sigmoid(lin("rgb_linear", h)), alpha], -1)


# ---------------------------------------------------------------------------
# a26 Network.render_rays / forward       network.py:1003-1042, 1092-1126
#     boost_mvsnerf Network               lib/networks/boost_mvsnerf/network.py:23-211
# ---------------------------------------------------------------------------

def point_inputs(rays, volume, src_inps, src_exts, src_ixts, near, far, Ns, capture=None):
    """rays (1,N,8) -> MLP input (N,Ns,86), z (1,N,Ns), xyz (1,N,Ns,3)."""
    H, W = src_inps.shape[-2:]
    xyz, z = ray_march(rays, Ns)
    inv_scale = torch.tensor([W - 1.0, H - 1.0])
    ndc = ndc_coordinate(src_exts[0, 0], src_ixts[0, 0], xyz[0], inv_scale, near, far, PAD)
    rgbs = E.unpreprocess(src_inps, 1.0)
    feat = torch.cat([volume_lookup(volume, ndc), colour_lookup(xyz[0], src_exts[0], src_ixts[0], rgbs)], -1)
    angle = dir_feature(src_exts[0, 0], rays[0, :, 3:6])
    x = torch.cat([embed(ndc), feat, angle[:, None].expand(-1, Ns, -1)], -1)
    if capture is not None:
        capture.update({"ndc": ndc, "feat": feat, "angle": angle, "xyz": xyz, "z": z})
    return x, z, xyz


def volume_for_views(sd, imgs, feats, exts, ixts, depth_ranges, D):
    dv, near, far = depth_planes(depth_ranges, D)
    vol = cost_volume(imgs, feats, proj_mats(exts, ixts), dv[None].expand(imgs.shape[0], -1))
    reg = cost_reg(sd, vol)
    return reg.reshape(1, -1, *reg.shape[2:]), near, far, vol


def mvsnerf_forward(sd, batch, cfg, capture=None):
    Ns = cfg.enerf.cas_config.num_samples[0]
    feats = feature_net(sd, batch["all_src_inps"])
    v = [0, 1, 2]
    imgs, exts, ixts = batch["all_src_inps"][:, v], batch["all_src_exts"][:, v], batch["all_src_ixts"][:, v]
    volume, near, far, raw_vol = volume_for_views(sd, imgs, feats[:, v], exts, ixts, batch["depth_ranges"][:, v], Ns)
    x, z, _ = point_inputs(batch["rays_0"], volume, imgs, exts, ixts, near, far, Ns, capture)
    raw = renderer_mlp(sd, x)[None]
    if capture is not None:
        capture.update({"cost_volume": raw_vol, "volume": volume, "mlp_in": x, "raw": raw})
    out = E.composite(raw, z, cfg.enerf.white_bkgd)
    return {k + "_level0": val for k, val in out.items()}


def boost_mvsnerf_forward(sd, batch, cfg, k_best, capture=None):
    Ns = cfg.enerf.cas_config.num_samples[0]
    N = batch["all_src_inps"].shape[1]
    trip = E.view_triplets(N, cfg.enerf.cost_volume_input_views)
    feats = feature_net(sd, batch["all_src_inps"])
    H, W = batch["all_src_inps"].shape[-2:]
    inv_scale = torch.tensor([[W - 1.0, H - 1.0]])
    raws, zs, ms = [], [], []
    for k in k_best:
        v = list(trip[k])
        imgs, exts, ixts = batch["all_src_inps"][:, v], batch["all_src_exts"][:, v], batch["all_src_ixts"][:, v]
        volume, near, far, _ = volume_for_views(sd, imgs, feats[:, v], exts, ixts, batch["depth_ranges"][:, v], Ns)
        x, z, xyz = point_inputs(batch["rays_0"], volume, imgs, exts, ixts, near, far, Ns)
        raws.append(renderer_mlp(sd, x)[None])
        zs.append(z)
        ms.append(E.viewport_mask(xyz, exts, ixts, inv_scale).reshape(1, -1, Ns))
    raws, zs, ms = torch.stack(raws, 1), torch.stack(zs, 1), torch.stack(ms, 1)
    if capture is not None:
        capture.update({"raws": raws, "zs": zs, "masks": ms})
    out = E.blend(raws, E.normalise_masks(ms), zs)
    return {k_ + "_level0": val for k_, val in out.items()}


def triplet_visibility(batch, cfg, ids, n_march=128):
    """calc_mask (boost_mvsnerf/network.py:23-45): no network involved, 128 marched samples."""
    ids = list(ids)
    H, W = batch["all_src_inps"].shape[-2:]
    xyz, z = ray_march(batch["rays_0"], n_march)
    inv_scale = torch.tensor([[W - 1.0, H - 1.0]])
    m = E.viewport_mask(xyz, batch["all_src_exts"][:, ids], batch["all_src_ixts"][:, ids], inv_scale)
    m = m.reshape(1, -1, n_march, 1) / n_march
    return E.composite(m.repeat(1, 1, 1, 4), z, cfg.enerf.white_bkgd)["rgb"].mean(-1)


def view_selection(batch, cfg):
    N = batch["all_src_inps"].shape[1]
    masks = [triplet_visibility(batch, cfg, ids) for ids in E.view_triplets(N, 3)]
    sel = E.greedy_cover(masks, cfg.enerf.cas_config.k_best)
    return {f"{batch['meta']['scene'][0]}_{batch['meta']['tar_view'][0]}": [int(s) for s in sel]}
