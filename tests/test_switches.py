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
