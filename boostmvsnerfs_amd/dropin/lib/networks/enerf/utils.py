# Drop-in for lib/networks/enerf/utils.py (functional API; star-imported by boost_mvsnerf).
from boostmvsnerfs_amd.networks.enerf.utils import *  # noqa: F401,F403
from boostmvsnerfs_amd.networks.enerf.utils import __all__  # noqa: F401
