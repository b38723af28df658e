"""bench.py's command line as the driver uses it (`--gpus N --steps K --warmup W`), its defaults, and the host-side
timer logic (`ktimer`) that needs no GPU."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _parse(mod, argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return mod.parse()
    finally:
        sys.argv = old


def test_driver_flags_and_defaults():
    b = _bench()
    a = _parse(b, ["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 5) and a.workload == b.HEADLINE
    assert a.spinup_steps == 100 and a.event_every == 10 and not a.pipelined and a.shard == "views"
    # no flags: N = 1; the ms-scale inference workloads average over many steps, the others over 30
    assert _parse(b, []).steps == 400 and _parse(b, []).gpus == 1
    assert _parse(b, ["--workload", "enerf_ours_480x736_6src_k4"]).steps == 400
    assert _parse(b, ["--workload", "enerf_ft_512x640_3src"]).steps == 30
    assert _parse(b, ["--workload", "mvsnerf_ours_224x352_128planes_k4"]).steps == 30
    # every BASELINE config has a workload
    assert len(b.WORKLOADS) >= 6 and b.WORKLOADS[b.HEADLINE]["planes"] == [64, 8]


def test_sweep_bytes_is_survey_8d():
    b = _bench()
    # config 2: level 0 (3 views x 32 ch x 128x160 source, 32 ch x 64 planes x 64x80 volume), level 1
    assert b.sweep_bytes(3, 32, 128, 160, 64, 64, 80) == 49807360
    assert b.sweep_bytes(3, 16, 256, 320, 8, 256, 320) == 57671680


def test_ktimer_is_inert_when_disabled():
    from boostmvsnerfs_amd import ktimer
    ktimer.reset()
    assert not ktimer.enabled
    with ktimer.region("sweep_variance[x]", bind=True):      # must not touch the library or the GPU
        pass
    ktimer.collect()
    assert ktimer.summary() == {}


def test_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` exactly as the driver types it for N = 1 (no torch.distributed.run in front): bench.py
    starts its two ranks itself, they rendezvous (gloo here, RCCL on the GPU box), run warm-up + K timed steps with the
    per-step exchange of the `views` sharding, and rank 0 prints ONE JSON line with n_gpus = 2.  --stub-renderer replaces
    the network (the HIP path cannot run on CPU); everything else is bench.main()."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "enerf_256x320_3src_32planes", "--stub-renderer"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["unit"] == "Mray/s" and line["value"] > 0 and "STUB" in line["data"]
    # whole-job aggregate: two frames per step
    assert abs(line["value"] - 2 * 256 * 320 * 3 / (line["ms_per_step"] * 3 * 1e-3) / 1e6) < 1e-6 * line["value"] + 1e-9
    assert line["config"]["shard"] == "views"
