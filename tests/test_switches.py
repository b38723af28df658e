"""The Python host's execution switches are explicit state (boostmvsnerfs_amd/switches.py): declared once, environment
applied once, unknown names refused; nothing else in the package reads os.environ for behaviour."""
import pathlib
import re

import pytest

from boostmvsnerfs_amd import switches


def test_defaults_override_and_unknown_names():
    assert switches.get("BMV_CNN") == "engine" and switches.on("BMV_CONV_C4")
    with switches.override(BMV_CNN="torch", BMV_CONV_C4=0):
        assert switches.get("BMV_CNN") == "torch" and not switches.on("BMV_CONV_C4")
    assert switches.get("BMV_CNN") == "engine" and switches.on("BMV_CONV_C4")
    with pytest.raises(KeyError):
        switches.get("BMV_NO_SUCH_SWITCH")
    with pytest.raises(KeyError):
        switches.set("BMV_NO_SUCH_SWITCH", 1)


def test_environment_is_parsed_once_and_loudly(monkeypatch):
    monkeypatch.setattr(switches, "VALUES", {})
    switches.apply_environment({"BMV_OVERLAP": " 1 ", "BMV_BOOST_STREAMS": "off", "BMV_CNN": "torch", "UNRELATED": "x"})
    assert switches.get("BMV_OVERLAP") == 1 and switches.get("BMV_BOOST_STREAMS") == 0 and switches.get("BMV_CNN") == "torch"
    with pytest.raises(ValueError):
        switches.apply_environment({"BMV_OVERLAP": "fast"})


def test_no_other_module_reads_the_environment_for_behaviour():
    root = pathlib.Path(switches.__file__).parent
    allowed = {"switches.py", "_lib.py", "build.py", "config.py"}     # config.py: the reference's own `workspace` variable
    offenders = [str(p.relative_to(root)) for p in root.rglob("*.py")
                 if p.name not in allowed and re.search(r"os\.environ|os\.getenv", p.read_text())]
    assert not offenders, offenders


def test_quad_volume_is_a_view_of_the_reference_layout():
    """ops.QuadVolume (B, C/4, D, h, w, 4): the layout the inference kernels hand cost volumes on in; to_planar() is the
    reference tensor, the wrapper answers the shape / dtype / device questions the modules ask of a tensor."""
    import torch
    from boostmvsnerfs_amd import ops
    x = torch.arange(2 * 8 * 3 * 4 * 5, dtype=torch.float32).view(2, 8, 3, 4, 5)
    q = ops.QuadVolume(x.view(2, 2, 4, 3, 4, 5).permute(0, 1, 3, 4, 5, 2).contiguous())
    assert q.shape == x.shape and q.dtype == x.dtype and q.device == x.device and not q.is_cuda
    assert torch.equal(q.to_planar(), x)
    with pytest.raises(ValueError):
        ops.QuadVolume(x)


def test_switches_set_after_import_are_honoured():
    """several switches were read once into module constants (convnet.CONV_C4 / SPLIT_BF16, cnn.FUSE_*,
    autograph.ENABLED / DEFER / RING / MAX_GRAPHS), so `switches.set()` / `override()` after import silently changed
    nothing.  They are read at construction / call time now; "08" parses as `_lib.load` parses the library's switches."""
    from boostmvsnerfs_amd import autograph, convnet
    from boostmvsnerfs_amd.networks.enerf import cnn
    for mod, names in ((convnet, ("CONV_C4", "SPLIT_BF16")), (cnn, ("FUSE_FPN_SMOOTH", "FUSE_CONV0", "FUSE_TOP")),
                       (autograph, ("ENABLED", "DEFER", "RING", "MAX_GRAPHS"))):
        for n in names:
            assert not hasattr(mod, n), f"{mod.__name__}.{n} is an import-time copy of a switch again"
    assert convnet.split_bf16_default() == 0
    with switches.override(BMV_CONV_SPLIT="auto"):
        assert convnet.split_bf16_default() == "auto"
    with switches.override(BMV_CONV_SPLIT="3", BMV_CONV_C4=0):
        assert convnet.split_bf16_default() == 3
        reg = cnn.MinCostRegNet(8)
        assert reg.split_bf16 == 3 and reg.conv_c4 is False
    assert cnn.MinCostRegNet(8).conv_c4 is True

    class _Net:
        training = False

        def parameters(self):
            return iter(())
    import torch
    ag = autograph.AutoGraph(_Net())
    with switches.override(BMV_AUTOGRAPH=0):
        assert ag.usable({"x": torch.zeros(1)}) is False
    assert switches._parse("BMV_AUTOGRAPH_MAX", "08") == 8
