"""BASELINE configs[2] and configs[3] at their own sizes (the CPU oracle cannot be the checker for a whole frame there):

* enerf_ours: 480x736 target, N = 6 source views, K = 4 cost volumes of 3 views, planes [64, 8]
  (configs/exps/evaluate/enerf_ours/free_eval.yaml; lib/networks/boost_enerf/network.py:172-237);
* mvsnerf_ours: 224x352, N = 6, K = 4, num_samples = [128] = 128 depth planes AND 128 samples per ray
  (configs/exps/evaluate/mvsnerf_ours/scannet_plus_eval.yaml + `enerf.cas_config.num_samples "[128]"`;
  lib/networks/boost_mvsnerf/network.py:160-211, mvsnerf/network.py:1094).

Size-independent properties: execution paths that must agree bit for bit (K streams == one stream, views by index ==
gathered views, ray shards == full frame), K = 1 fusion == the single-volume network, invariance of the fused picture to
the order of the volumes, and the 128-sample MLP / sampler against the CPU oracle on a ray subset."""
import json

import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _write_sel(tmp, sel):
    with open(tmp / "view_selection.json", "w") as f:
        json.dump({"synthetic_0": sel}, f)


# ---------------------------------------------------------------------------------------------------- config 3
@pytest.fixture(scope="module")
def cfg3(tmp_path_factory):
    from boostmvsnerfs_amd.config import get_cfg, make_cfg, set_cfg
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    prev = get_cfg()
    tmp = tmp_path_factory.mktemp("cfg3")
    cfg = make_cfg("enerf_ours_eval")
    cfg.enerf.cas_config.volume_planes = [64, 8]
    cfg.enerf.cas_config.k_best = 4
    cfg.result_dir = str(tmp)
    set_cfg(cfg)
    batch = clone_batch(make_batch(480, 736, n_views=6, seed=0), DEV)
    yield cfg, tmp, batch
    set_cfg(prev)


def _boost_enerf(tmp, sel, seed=0):
    from boostmvsnerfs_amd.networks.boost_enerf.network import Network
    _write_sel(tmp, sel)
    torch.manual_seed(seed)
    return Network().eval().to(DEV)


def _run_raw(net, batch):
    from boostmvsnerfs_amd.synthetic import clone_batch
    with torch.no_grad():
        out = net(clone_batch(batch))
    return {k: (tuple(t.clone() for t in v) if isinstance(v, tuple) else v.clone()) for k, v in out.items()}


def _run(net, batch):
    from boostmvsnerfs_amd.synthetic import clone_batch
    with torch.no_grad():
        return {k: v.clone() for k, v in net(clone_batch(batch)).items()}


def test_config3_execution_paths_agree_bit_for_bit(cfg3):
    cfg, tmp, batch = cfg3
    assert batch["all_src_inps"].shape == (1, 6, 3, 480, 736)
    net = _boost_enerf(tmp, [0, 7, 13, 19])
    want = _run(net, batch)                                  # default: K streams, views by index
    assert len(net._streams) == 4
    assert want["rgb_level1"].shape == (1, 480 * 736, 3) and bool(torch.isfinite(want["rgb_level1"]).all())
    net.batched_volumes = True
    batched = _run(net, batch)                               # the K volumes as one batch through the regularisers
    net.batched_volumes = False
    from conftest import assert_close
    for k in want:   # (batch K picks other convolution tilings than batch 1: same values up to summation order)
        assert_close(batched[k], want[k], rtol=1e-4, atol_scale=1e-4, name=f"{k}: one batch of K volumes vs K streams", max_outlier_frac=2e-3)
    net.parallel_volumes = False
    seq = _run(net, batch)
    net.by_index = False
    gathered = _run(net, batch)
    for k in want:
        assert torch.equal(seq[k], want[k]), f"{k}: one stream vs K streams"
        assert torch.equal(gathered[k], want[k]), f"{k}: gathered views vs views by index"


def test_config3_volume_subsets_and_ray_ranges_reassemble(cfg3):
    """What `bench.py --shard volumes` (sharding.VolumeShard) asks of each rank: a subset of the K cost volumes over a
    ray range, returned as stacked (raw, z, mask).  Two ranks' worth of subsets, interleaved back and fused, are the
    full frame bit for bit; a ray range of the fused path equals that slice of the full frame."""
    cfg, tmp, batch = cfg3
    net = _boost_enerf(tmp, [0, 7, 13, 19])
    want = _run(net, batch)
    N = 480 * 736
    rng = (N // 3 + 5, N // 3 + 5 + 100_003)
    net.ray_range = rng
    part = _run(net, batch)
    for k in ("rgb_level1", "depth_level1", "weights_level1"):
        assert torch.equal(part[k], want[k][:, rng[0]:rng[1]]), k
    halves = []
    for ids in ([0, 2], [1, 3]):
        net.volume_ids = ids
        out = _run_raw(net, batch)
        assert ("depth_mvs_level1" in out) == (0 in ids)             # depth_mvs / std belong to volume 0's owner
        halves.append(out["stacks_level1"])
    net.volume_ids = None
    net.ray_range = None
    raws, zs, ms = (torch.stack([halves[0][j][:, 0], halves[1][j][:, 0], halves[0][j][:, 1], halves[1][j][:, 1]], 1)
                    for j in range(3))
    assert raws.shape == (1, 4, rng[1] - rng[0], 2, 4)
    fused = net.merge_mlp_outputs(raws.contiguous(), ms.contiguous(), zs.contiguous())
    assert torch.equal(fused["rgb"], want["rgb_level1"][:, rng[0]:rng[1]])
    assert torch.equal(fused["depth"], want["depth_level1"][:, rng[0]:rng[1]])


def test_config3_volume_order_does_not_change_the_fused_picture(cfg3):
    cfg, tmp, batch = cfg3
    a = _run(_boost_enerf(tmp, [0, 7, 13, 19]), batch)
    b = _run(_boost_enerf(tmp, [19, 13, 0, 7]), batch)
    for k in ("rgb_level1", "depth_level1", "weights_level1"):       # depth_mvs / std come from volume 0: they change
        assert float((a[k] - b[k]).abs().max()) <= 1e-5 * max(1.0, float(a[k].abs().max())), k
    assert not torch.equal(a["depth_mvs_level1"], b["depth_mvs_level1"])


def test_config3_one_volume_is_plain_enerf(cfg3):
    """k_best = 1: masks normalise to 1 (or the 1/K = 1 fallback), so the fusion is ENeRF's compositing of that
    triplet up to the 1e-10 the single-volume transmittance carries (enerf/utils.py:618 vs :650)."""
    from boostmvsnerfs_amd.networks.boost_enerf.network import view_triplets
    from boostmvsnerfs_amd.networks.enerf.network import Network as ENeRF
    from boostmvsnerfs_amd.synthetic import clone_batch
    cfg, tmp, batch = cfg3
    cfg.enerf.cas_config.k_best = 1
    try:
        t = 7
        net = _boost_enerf(tmp, [t])
        got = _run(net, batch)
        ids = list(view_triplets(6, 3)[t])
        torch.manual_seed(0)
        plain = ENeRF().eval().to(DEV)
        plain.load_state_dict(net.state_dict(), strict=True)
        b = clone_batch(batch)
        for k in ("src_inps", "src_exts", "src_ixts"):
            b[k] = batch["all_" + k][:, ids].contiguous()
        with torch.no_grad():
            want = plain(b)
    finally:
        cfg.enerf.cas_config.k_best = 4
    for k in ("rgb_level1", "depth_mvs_level1", "std_level1"):
        assert_close(got[k], want[k], rtol=1e-5, atol_scale=1e-5, name=k)
    assert_close(got["depth_level1"], want["depth_level1"], rtol=1e-4, atol_scale=1e-5, name="depth_level1")


# ---------------------------------------------------------------------------------------------------- config 4
@pytest.fixture(scope="module")
def cfg4(tmp_path_factory):
    from boostmvsnerfs_amd.config import get_cfg, make_cfg, set_cfg
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    prev = get_cfg()
    tmp = tmp_path_factory.mktemp("cfg4")
    cfg = make_cfg("mvsnerf_ours_eval")
    cfg.enerf.cas_config.num_samples = [128]
    cfg.enerf.cas_config.k_best = 4
    cfg.result_dir = str(tmp)
    set_cfg(cfg)
    b = make_batch(224, 352, n_views=6, seed=0, depth_ranges=True, render_scales=(1.0,))
    b["rays_0"][..., 6], b["rays_0"][..., 7] = 2.2, 7.5       # a real depth interval in the near / far columns (quirk 9)
    yield cfg, tmp, clone_batch(b, DEV)
    set_cfg(prev)


def _boost_mvs(tmp, sel):
    from boostmvsnerfs_amd.networks.boost_mvsnerf.network import Network
    _write_sel(tmp, sel)
    torch.manual_seed(0)
    return Network().eval().to(DEV)


def test_config4_ray_shards_equal_the_full_frame(cfg4):
    """`ray_range` of the MVS path (how configs[3] is sharded over 8 GPUs): ragged shards, bit-equal to the full frame."""
    cfg, tmp, batch = cfg4
    net = _boost_mvs(tmp, [0, 7, 13, 19])
    N = 224 * 352
    whole = _run(net, batch)
    assert whole["rgb_level0"].shape == (1, N, 3) and whole["weights_level0"].shape == (1, N, 128)
    assert bool(torch.isfinite(whole["rgb_level0"]).all())
    cuts = [0, N // 8, N // 8 + 10_001, N // 2 + 37, N]
    parts = []
    for b, e in zip(cuts[:-1], cuts[1:]):
        net.ray_range = (b, e)
        parts.append(_run(net, batch))
    net.ray_range = None
    for k in whole:
        got = torch.cat([p[k] for p in parts], 1)
        assert torch.equal(got, whole[k]), k
    # the 8-way split bench.py --shard rays uses
    from boostmvsnerfs_amd import sharding
    net.ray_range = sharding.ray_slice(N, 8, 5)
    part = _run(net, batch)
    b, e = net.ray_range
    net.ray_range = None
    assert torch.equal(part["rgb_level0"], whole["rgb_level0"][:, b:e])


def test_config4_blend_invariants(cfg4):
    cfg, tmp, batch = cfg4
    a = _run(_boost_mvs(tmp, [0, 7, 13, 19]), batch)
    b = _run(_boost_mvs(tmp, [13, 0, 19, 7]), batch)
    for k in a:
        assert float((a[k] - b[k]).abs().max()) <= 2e-5 * max(1.0, float(a[k].abs().max())), k
    w = a["weights_level0"]
    assert float(w.min()) >= 0 and float((w.sum(-1) - 1).abs().max()) <= 1e-4          # softmaxed weights (quirk 1)
    assert float(a["rgb_level0"].min()) >= -1e-6


def test_config4_sampler_and_mlp_at_128_samples_vs_oracle(cfg4):
    """One 128-plane cost volume of the frame; 384 rays of it through the fused sampler + 6x128 MLP against
    oracle/mvsnerf.py (ray march, NDC, embedding, lookups, Renderer_ours) at Ns = 128."""
    from boostmvsnerfs_amd import ops
    from oracle import mvsnerf as M
    cfg, tmp, batch = cfg4
    from boostmvsnerfs_amd.synthetic import clone_batch
    net = _boost_mvs(tmp, [0, 7, 13, 19])
    b = clone_batch(batch)
    with torch.no_grad():
        feats = net.feature(b["all_src_inps"])
        ids = torch.tensor([[1, 3, 4]], device=DEV)
        st = net.build_volume(b, feats, ids)
    assert st.volume.shape[1] == 128
    rays = torch.cat([b["rays_0"][0, :128], b["rays_0"][0, 40_000:40_128], b["rays_0"][0, -128:]]).contiguous()
    raw, z, mask, x86 = ops.mvs_render(rays, st.volume, *st.views, st.near_far, net.nerf.nerf.packed_weights(), Ns=128,
                                       pad=24, want_mask=True, want_inputs=True)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    near, far = float(st.near_far[0]), float(st.near_far[1])
    want_x, want_z, _ = M.point_inputs(rays.cpu()[None], st.volume.cpu()[None], *(t.cpu()[None] for t in st.views), near, far, 128)
    assert_close(z, want_z.reshape(z.shape), rtol=1e-5, atol_scale=1e-6, name="z")
    want_x = want_x.reshape(x86.shape)
    assert_close(x86[..., 63:], want_x[..., 63:], name="features + view direction", max_outlier_frac=1e-3)
    assert_close(x86[..., :3], want_x[..., :3], rtol=1e-4, atol_scale=1e-5, name="ndc")
    want_raw = M.renderer_mlp(sd, x86.cpu()).reshape(raw.shape)       # the MLP on the kernel's own inputs
    assert_close(raw, want_raw, name="raw (Ns = 128)")
