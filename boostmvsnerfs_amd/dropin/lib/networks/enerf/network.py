# Drop-in replacement for the reference's lib/networks/enerf/network.py: the reference's
# make_network() (lib/networks/make_network.py:3-10) loads this file by path and calls Network().
from boostmvsnerfs_amd.networks.enerf.network import Network  # noqa: F401
