"""Network factory with the contract of lib/networks/make_network.py:3-10:
`cfg.network_module` names the module, its `Network([preprocess])` is returned."""
import importlib

_ALIASES = {
    "lib.networks.enerf.network": "boostmvsnerfs_amd.networks.enerf.network",
    "lib.networks.boost_enerf.network": "boostmvsnerfs_amd.networks.boost_enerf.network",
    "lib.networks.mvsnerf.network": "boostmvsnerfs_amd.networks.mvsnerf.network",
    "lib.networks.boost_mvsnerf.network": "boostmvsnerfs_amd.networks.boost_mvsnerf.network",
}


def make_network(cfg, preprocess=False):
    name = cfg.network_module
    mod = importlib.import_module(_ALIASES.get(name, name))
    return mod.Network(preprocess) if preprocess else mod.Network()
