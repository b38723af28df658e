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
    assert a.spinup_steps == 100 and a.event_every == 4 and not a.pipelined and a.shard == "views"
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
