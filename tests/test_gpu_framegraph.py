"""HIP-graph replay of a frame (boostmvsnerfs_amd/framegraph.py) returns exactly what the eager forward returns,
for the single-volume network (two graphs cut at the level-1 sweep) and for the K-volume boost network (K streams
forked and joined inside one graph); refreshing the static batch in place changes the replayed result accordingly."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _eager(net, batch):
    with torch.no_grad():
        return {k: v.clone() for k, v in net(batch).items()}


def test_enerf_replay_equals_eager():
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.framegraph import FrameGraph
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [16, 8]
    set_cfg(cfg)
    torch.manual_seed(0)
    net = Network().eval().to(DEV)
    batch = clone_batch(make_batch(128, 160, n_views=3, seed=0), DEV)
    want = _eager(net, batch)
    fg = FrameGraph(net, batch)
    assert len(fg.graphs) == 2 and fg.sweep_args is not None          # cut at the level-1 sweep
    for _ in range(3):
        got = fg.replay()
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # new frame: refresh the static inputs in place, replay, compare with an eager pass on the same data
    other = clone_batch(make_batch(128, 160, n_views=3, seed=0, tar_offset=(0.1, 0.0, 0.0)), DEV)
    for k, v in other.items():
        if torch.is_tensor(v):
            batch[k].copy_(v)
    got = fg.replay()
    torch.cuda.synchronize()
    want2 = _eager(net, other)
    assert not torch.equal(want2["rgb_level1"], want["rgb_level1"])
    for k in want2:
        assert torch.equal(got[k], want2[k]), k


def test_boost_replay_equals_eager(tmp_path):
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.framegraph import FrameGraph
    from boostmvsnerfs_amd.networks.boost_enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    cfg = make_cfg("enerf_ours_eval")
    cfg.enerf.cas_config.volume_planes = [16, 8]
    cfg.enerf.cas_config.k_best = 3
    cfg.result_dir = str(tmp_path)
    set_cfg(cfg)
    batch = clone_batch(make_batch(128, 160, n_views=5, seed=0), DEV)
    key = f"{batch['meta']['scene'][0]}_{batch['meta']['tar_view'][0]}"
    with open(tmp_path / "view_selection.json", "w") as f:
        json.dump({key: [0, 4, 7]}, f)
    torch.manual_seed(0)
    net = Network().eval().to(DEV)
    want = _eager(net, batch)
    fg = FrameGraph(net, batch)
    for _ in range(2):
        got = fg.replay()
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_event_brackets_inside_a_graph():
    """FrameGraph(events=True): the sweeps and the renderer are bracketed by event-record nodes of the ONE graph
    (csrc/timing.hip; bench.py's roofline durations).  The replay returns the eager result, every replay re-stamps the
    brackets, and a bracket reads a plausible duration: positive, and no longer than the whole replay."""
    import time

    from boostmvsnerfs_amd import ktimer
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.framegraph import FrameGraph
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [16, 8]
    set_cfg(cfg)
    torch.manual_seed(0)
    net = Network().eval().to(DEV)
    batch = clone_batch(make_batch(128, 160, n_views=3, seed=0), DEV)
    want = _eager(net, batch)
    ktimer.reset()
    ktimer.forget_graph_events()
    ktimer.only = ("sweep_variance", "render_rays")
    try:
        fg = FrameGraph(net, batch, cut=None, events=True)
        assert len(fg.graphs) == 1 and not fg.sweeps
        ktimer.enabled = True
        walls = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = fg.replay()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
            ktimer.collect()
        ks = ktimer.summary()
    finally:
        ktimer.enabled, ktimer.only = False, None
        ktimer.forget_graph_events()
    for k in want:
        assert torch.equal(got[k], want[k]), k
    sweeps = {n: v for n, v in ks.items() if n.startswith("sweep_variance")}
    render = {n: v for n, v in ks.items() if n.startswith("render_rays")}
    assert len(sweeps) == 2 and len(render) == 1, ks
    for name, (launches, mean_ms, min_ms) in ks.items():
        assert launches == 3, (name, launches)
        assert 0.0 < min_ms <= mean_ms < max(walls), (name, mean_ms, walls)


def test_bound_events_on_the_sweeps_between_graphs():
    """FrameGraph(cut="all", events=True), bench.py's bracketed frame: the sweeps are ordinary launches between the
    graphs with events BOUND to their dispatch (bmv_bind_next_launch -> hipExtLaunchKernelGGL), the renderer keeps its
    in-graph bracket; results equal the eager ones and every bracket reads a plausible duration."""
    from boostmvsnerfs_amd import ktimer
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.framegraph import FrameGraph
    from boostmvsnerfs_amd.networks.enerf.network import Network
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [16, 8]
    set_cfg(cfg)
    torch.manual_seed(0)
    net = Network().eval().to(DEV)
    batch = clone_batch(make_batch(128, 160, n_views=3, seed=0), DEV)
    want = _eager(net, batch)
    ktimer.reset()
    ktimer.forget_graph_events()
    ktimer.only = ("sweep_variance", "render_rays")
    try:
        fg = FrameGraph(net, batch, cut="all", events=True)
        assert len(fg.sweeps) == 2 and len(fg.graphs) == 3
        ktimer.reset()
        ktimer.enabled = True
        for _ in range(3):
            got = fg.replay()
            torch.cuda.synchronize()
            ktimer.collect()
        ks = ktimer.summary()
        assert not ktimer._no_bind                       # the windowed sweep took the bound events
    finally:
        ktimer.enabled, ktimer.only = False, None
        ktimer.forget_graph_events()
        ktimer.reset()
    for k in want:
        assert torch.equal(got[k], want[k]), k
    sweeps = {n: v for n, v in ks.items() if n.startswith("sweep_variance")}
    render = {n: v for n, v in ks.items() if n.startswith("render_rays")}
    assert len(sweeps) == 2 and len(render) == 1, ks
    for name, (launches, mean_ms, min_ms) in ks.items():
        assert launches == 3 and 0.0 < min_ms <= mean_ms < 5.0, (name, launches, mean_ms)


def _small_net():
    from boostmvsnerfs_amd.config import make_cfg, set_cfg
    from boostmvsnerfs_amd.networks.enerf.network import Network
    cfg = make_cfg("enerf_eval")
    cfg.enerf.cas_config.volume_planes = [16, 8]
    set_cfg(cfg)
    torch.manual_seed(0)
    net = Network().eval().to(DEV)

    def eager(b):
        with torch.no_grad():
            return {k: v.clone() for k, v in net._forward_checked(dict(b)).items()}
    return net, eager


def test_forward_captures_itself():
    """autograph.AutoGraph behind Network.forward: the drop-in call replays a HIP graph from the second call with the
    same shapes on.  Default contract = the reference's forward: inputs are never written, outputs are fresh tensors
    (captured on private copies of the batch); a parameter update re-captures, another shape falls back to an eager
    first call, a batch tensor that requires grad keeps the call eager."""
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    net, eager = _small_net()
    ag = net._autograph

    def snapshot(b):
        return {k: v.clone() for k, v in b.items() if torch.is_tensor(v)}

    def unchanged(b, snap):
        return all(torch.equal(b[k], v) for k, v in snap.items())

    batch = clone_batch(make_batch(128, 160, n_views=3, seed=0), DEV)
    other = clone_batch(make_batch(128, 160, n_views=3, seed=1), DEV)
    snap_a, snap_b = snapshot(batch), snapshot(other)
    want, want_other = eager(batch), eager(other)
    with torch.no_grad():
        outs = [net(batch) for _ in range(4)]          # NOT cloned: every call must hand out its own tensors
    torch.cuda.synchronize()
    assert ag.stats["eager"] == 1 and ag.stats["captures"] == 1 and ag.stats["replays"] == 3 and ag.stats["copies"] > 0
    for o in outs:
        for k in want:
            assert torch.equal(o[k], want[k]), k
    assert len({o["rgb_level1"].data_ptr() for o in outs}) == 4
    # the images and the rays are READ IN PLACE and the renderer WRITES the returned tensors (pointer table,
    # autograph._adopt_table): checked once at capture, nothing rejected, no copy of those tensors
    assert ag.stats.get("defer_checks") == 1 and "defer_rejected" not in ag.stats and ag.stats.get("deferred", 0) > 0
    d = ag._hot["defer"]
    assert {k for k, _ in d["in"]} == {"src_inps", "rays_1"}
    # ... rgb / depth / weights written by the renderer, the small maps copied by nodes of the frame's own graph
    assert {k for k, _, _ in d["out"]} == {"rgb_level1", "depth_level1", "weights_level1", "depth_mvs_level1", "std_level1"}
    assert "src_inps" not in ag._hot["names"] and "tar_ext" in ag._hot["names"]
    assert d["feed"] is not None and d["feed"]["m"] == len(ag._hot["names"])
    # ... and no launch in front of the replay: the frame's first node reads this call's message from a host ring; every
    # replay found the message with its own sequence number
    ring = ag._hot["ring"]
    assert ring is not None and ring.faults() == 0 and int(ring.state[0].item()) == ring.posted > 3
    # two batches of the same shapes, alternated twice (with the graph captured on the caller's tensors the
    # second pass rendered the wrong frame and the first batch's tensors had been overwritten)
    with torch.no_grad():
        frames = [net(b) for b in (batch, other, batch, other)]
    torch.cuda.synchronize()
    assert ag.stats["captures"] == 1
    for f, w in zip(frames, (want, want_other, want, want_other)):
        for k in w:
            assert torch.equal(f[k], w[k]), k
    assert unchanged(batch, snap_a) and unchanged(other, snap_b)
    assert not torch.equal(want_other["rgb_level1"], want["rgb_level1"])
    assert ring.faults() == 0 and int(ring.state[0].item()) == ring.posted
    # more frames than the ring has slots, WITHOUT a synchronize per frame (the host runs ahead of the GPU: a slot must
    # not be rewritten before the replay that reads it has run -- FeedRing._reserve)
    with torch.no_grad():
        for i in range(3 * ring.R + 5):
            f = net(other if i % 2 else batch)
    torch.cuda.synchronize()
    assert ring.faults() == 0 and int(ring.state[0].item()) == ring.posted
    for k in want:
        assert torch.equal(f[k], want[k]), k
    # a message that is posted and never replayed (an exception between the two) is superseded by the next post
    ring.post()
    with torch.no_grad():
        f = net(other)
    torch.cuda.synchronize()
    assert ring.faults() == 0 and int(ring.state[0].item()) == ring.posted
    for k in want_other:
        assert torch.equal(f[k], want_other[k]), k
    # a parameter update invalidates the captured frame
    with torch.no_grad():
        next(net.nerf_1.parameters()).mul_(1.01)
    want4 = eager(other)
    with torch.no_grad():
        got4 = net(other)
    assert ag.stats["captures"] == 2
    for k in want4:
        assert torch.equal(got4[k], want4[k]), k
    # load_state_dict invalidates too (assign=True replaces the Parameter objects the version check walks)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    first = next(iter(k for k in sd if k.startswith("nerf_1") and sd[k].is_floating_point()))
    sd[first] = sd[first] * 1.01
    net.load_state_dict(sd, assign=True)
    want4b = eager(other)
    assert not torch.equal(want4b["rgb_level1"], want4["rgb_level1"])
    with torch.no_grad():
        for _ in range(3):
            got4b = net(other)
    for k in want4b:
        assert torch.equal(got4b[k], want4b[k]), k
    # another size: first call eager
    small = clone_batch(make_batch(64, 96, n_views=3, seed=0), DEV)
    n_eager = ag.stats["eager"]
    with torch.no_grad():
        got5 = net(small)
    assert ag.stats["eager"] == n_eager + 1
    want5 = eager(small)
    for k in want5:
        assert torch.equal(got5[k], want5[k]), k
    # a batch tensor that requires grad (pose refinement with frozen weights): eager, the output carries the graph
    for p in net.parameters():
        p.requires_grad_(False)
    posed = dict(other)
    posed["tar_ext"] = other["tar_ext"].clone().requires_grad_(True)
    assert not ag.usable(posed) and ag.usable(other)
    for p in net.parameters():
        p.requires_grad_(True)
    # training / autograd never replays
    net.train()
    assert not ag.usable(batch)
    net.eval()


def test_pipelined_caller_alternating_contiguous_and_strided_small_inputs():
    """a caller that does NOT synchronise per frame hands `tar_ext` alternately as a strided view
    (the slow post(): private copy + a message with n_copy = 0) and as a contiguous tensor (the fast path, whose
    prepare_fast used to rewrite n_copy / sources in ALL ring messages, also the queued slow one the GPU had not
    consumed yet -> that frame's first node overwrote the just-copied camera with stale sources).  Every frame must be
    the eager frame of its own camera."""
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    net, eager = _small_net()
    ag = net._autograph
    cams = [clone_batch(make_batch(128, 160, n_views=3, seed=0, tar_offset=(0.02 * i, 0.0, 0.0)), DEV) for i in range(4)]
    wants = [eager(b) for b in cams]
    assert not torch.equal(wants[0]["rgb_level1"], wants[1]["rgb_level1"])

    def strided(b):
        c = dict(b)
        wide = torch.zeros(1, 4, 8, device=DEV)
        wide[..., ::2] = b["tar_ext"]
        c["tar_ext"] = wide[..., ::2]                   # same values, not contiguous
        assert not c["tar_ext"].is_contiguous()
        return c
    with torch.no_grad():
        for _ in range(3):
            net(cams[0])                                # eager, capture, first replay
        torch.cuda.synchronize()
        frames = []
        for i in range(40):                             # no synchronize inside: the host runs ahead of the GPU
            b = cams[i % 4]
            frames.append((net(strided(b) if i % 2 else b), wants[i % 4]))
    torch.cuda.synchronize()
    assert ag.stats["captures"] == 1
    for got, want in frames:
        for k in want:
            assert torch.equal(got[k], want[k]), k
    ring = ag._hot["ring"]
    assert ring.faults() == 0 and int(ring.state[0].item()) == ring.posted
    ag.check_faults()


def test_two_captured_frames_alternate():
    """Two keys alive at once (an execution switch flipped between calls; for the K-volume networks: two targets with
    different triplets): the steady-state path of one entry must hand over to the other without leaving a ring message
    un-replayed (every execution of a frame's first node reads the message with its own number)."""
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    net, eager = _small_net()
    ag = net._autograph
    a = clone_batch(make_batch(128, 160, n_views=3, seed=0), DEV)
    b = clone_batch(make_batch(128, 160, n_views=3, seed=1), DEV)
    want_a, want_b = eager(a), eager(b)
    with torch.no_grad():
        for _ in range(3):
            net(a)
        net.frame_setup = False          # another key, the same picture
        for _ in range(3):
            net(b)
        assert ag.stats["captures"] == 2
        frames = []
        for i in range(8):
            net.frame_setup = i % 2 == 0
            frames.append((net(a if i % 3 else b), want_a if i % 3 else want_b))
    torch.cuda.synchronize()
    assert ag.stats["captures"] == 2
    for got, want in frames:
        for k in want:
            assert torch.equal(got[k], want[k]), k
    for e in ag.entries.values():
        ring = e["ring"]
        assert ring.faults() == 0 and int(ring.state[0].item()) == ring.posted


def test_deferral_is_rejected_when_a_kernel_reads_the_captured_copy():
    """With the renderer's lookup records off, the fused renderer reads the source colours straight from `src_inps`
    (a baked pointer): the capture-time check must notice that the frame does not read the images through the table
    alone and keep the copies -- and the frames must still be right."""
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    net, eager = _small_net()
    net.lookup_records = False
    ag = net._autograph
    a = clone_batch(make_batch(128, 160, n_views=3, seed=0), DEV)
    b = clone_batch(make_batch(128, 160, n_views=3, seed=1), DEV)
    want_a, want_b = eager(a), eager(b)
    with torch.no_grad():
        frames = [net(x) for x in (a, a, b, a, b)]
    torch.cuda.synchronize()
    assert ag.stats["captures"] == 1 and ag.stats.get("defer_checks") == 1 and ag.stats.get("defer_rejected") == 1
    assert ag._hot["defer"] is None and "src_inps" in ag._hot["names"]
    for f, w in zip(frames, (want_a, want_a, want_b, want_a, want_b)):
        for k in w:
            assert torch.equal(f[k], w[k]), k


def test_deferred_pointer_nobody_takes_fails_the_launch():
    """include/bmv.h: a pointer registered with bmv_defer_pointer that the next launch does not read through a table
    fails that call (never a silently baked pointer)."""
    from boostmvsnerfs_amd import _lib, ops
    lib = _lib.load()
    table = torch.zeros(16, dtype=torch.int64, device=DEV)
    x = torch.rand(1, 2, device=DEV)
    assert lib.bmv_defer_pointer(x.data_ptr(), table.data_ptr(), 0) == 0
    with pytest.raises(RuntimeError, match="bmv_defer_pointer"):
        ops.depth_values_uniform(x, 4, 8, 8, True)
    assert lib.bmv_deferred_pending() == 0          # ... and the registration is gone
    ops.depth_values_uniform(x, 4, 8, 8, True)
    # the table setter
    vals = [x.data_ptr(), table.data_ptr()]
    tb = ops.PtrTable(torch.device(DEV))
    tb.set([3, 5], vals)
    torch.cuda.synchronize()
    assert tb.t[3].item() == vals[0] and tb.t[5].item() == vals[1] and tb.t[0].item() == 0


def test_forward_resident_opt_in():
    """`net.resident_inputs = True` / `net.alias_outputs = True` (autograph.py): the caller declares its batch resident
    -- captured on its own tensors, nothing copied per frame, in-place edits picked up; outputs are the graph's static
    tensors."""
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    net, eager = _small_net()
    net.resident_inputs = True
    net.alias_outputs = True
    ag = net._autograph
    batch = clone_batch(make_batch(128, 160, n_views=3, seed=0), DEV)
    want = eager(batch)
    with torch.no_grad():
        outs = [net(batch) for _ in range(4)]
        for k in want:
            assert torch.equal(outs[-1][k], want[k]), k
    assert ag.stats["eager"] == 1 and ag.stats["captures"] == 1 and ag.stats["replays"] == 3 and ag.stats["copies"] == 0
    assert outs[-1]["rgb_level1"].data_ptr() == outs[-2]["rgb_level1"].data_ptr()
    # the caller moves the camera IN PLACE in its resident batch: the replay reads the same tensors
    batch["tar_ext"][..., 0, 3] += 0.05
    want2 = eager(batch)
    with torch.no_grad():
        got2 = net(batch)
    assert ag.stats["captures"] == 1 and ag.stats["copies"] == 0
    assert not torch.equal(want2["rgb_level1"], want["rgb_level1"])
    for k in want2:
        assert torch.equal(got2[k], want2[k]), k
    # a new batch (other tensors, same shapes): copied into the declared batch's tensors, same graph
    other = clone_batch(make_batch(128, 160, n_views=3, seed=1), DEV)
    want3 = eager(other)
    with torch.no_grad():
        got3 = net(other)
    assert ag.stats["captures"] == 1 and ag.stats["copies"] > 0
    for k in want3:
        assert torch.equal(got3[k], want3[k]), k


def test_run_py_loop_with_new_device_tensors_every_frame_stays_on_the_fast_path():
    """run.py:113-123 as the reference runs it: every frame the loader's HOST batch is moved into NEW device tensors
    (`batch[k] = batch[k].cuda()`) and handed to `network(batch)`.  VERDICT r5 item 4 (bench's `host_batch_sync` read
    2.6x slower than the eager leg): the probe (profiles/r6/host_batch_sync_probe.txt) shows one eager + one capturing
    call and then nothing but replays fed through the host ring's fast path -- asserted here by the counters: 20 frames
    on never-seen addresses = 1 eager call, 1 capture, 18 replays, every replay through `post_fast` (no copy launch, no
    slow `post`), no ring fault, and every frame bit-equal to the eager frame of the same host batch."""
    from boostmvsnerfs_amd.synthetic import clone_batch, make_batch
    net, eager = _small_net()
    ag = net._autograph
    hosts = [make_batch(128, 160, n_views=3, seed=s) for s in (0, 1)]
    reads = net._autograph_inputs(clone_batch(hosts[0], DEV))
    pinned = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in h.items()
               if not torch.is_tensor(v) or reads is None or k in reads} for h in hosts]
    want = [eager(clone_batch(h, DEV)) for h in pinned]
    seen_ptrs = set()
    slow_posts = []
    frames = []
    for i in range(20):
        fresh = {k: (v.to(DEV, non_blocking=True) if torch.is_tensor(v) else v) for k, v in pinned[i % 2].items()}
        seen_ptrs.add(fresh["src_inps"].data_ptr())
        with torch.no_grad():
            out = net(fresh)
        torch.cuda.synchronize()                     # run.py's bracket
        frames.append(out)
        if i == 1:                                   # from the capture on: the slow post must never run again
            ring = ag._hot["ring"]
            real_post = ring.post

            def post(*a, **k):
                slow_posts.append(a)
                return real_post(*a, **k)
            ring.post = post
    assert ag.stats["eager"] == 1 and ag.stats["captures"] == 1 and ag.stats["replays"] == 19, ag.stats
    assert "defer_rejected" not in ag.stats
    ring = ag._hot["ring"]
    assert ring.fast is not None and not slow_posts
    assert ring.faults() == 0 and int(ring.state[0].item()) == ring.posted
    for i, f in enumerate(frames):
        for k in want[i % 2]:
            assert torch.equal(f[k], want[i % 2][k]), (i, k)
    assert len({f["rgb_level1"].data_ptr() for f in frames[2:]}) > 1           # (outputs are not one recycled buffer)
