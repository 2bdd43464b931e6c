"""Top-level `pygicp` module, as installed by the reference (src/python/main.cpp), served by the MI355X engine."""
from fast_gicp_amd.pygicp import *  # noqa: F401,F403
from fast_gicp_amd.pygicp import __version__  # noqa: F401
from fast_gicp_amd.pygicp import _kdtree_knn  # noqa: F401  (testing hook)
