# Same contract as the reference's factory (imp.load_source was removed in Python 3.12; importlib does the same).
import importlib.util


def make_network(cfg, preprocess=False):
    spec = importlib.util.spec_from_file_location(cfg.network_module, cfg.network_path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Network(preprocess) if preprocess else mod.Network()
