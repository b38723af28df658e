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
