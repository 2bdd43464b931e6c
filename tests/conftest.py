import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7, but libtorch_hip asks for "libamdhip64.so"), so a
# process that loads libfast_vgicp_hip.so FIRST ends up with two HIP runtimes and torch then sees no GPU. Loading torch
# first makes the engine bind to torch's copy (same SONAME). Tests that hand torch device pointers to the engine need
# that, so torch goes in before anything can load the engine. ("Sharing a process with PyTorch".)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing for device-pointer tests only
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from fast_gicp_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the GPU tests instead of failing them with 'create: status 3'
    (the engine has no CPU fallback); `-m gpu` on the GPU box still runs everything."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items or _has_gpu():
        return
    markexpr = (config.getoption("-m") or "").strip()
    if markexpr == "gpu" or os.environ.get("FVH_REQUIRE_GPU") == "1":
        return  # the GPU run was asked for explicitly: a missing device / library must fail loudly, never skip
    skip = pytest.mark.skip(reason="no MI355X visible (the HIP engine has no CPU fallback)")
    for it in gpu_items:
        it.add_marker(skip)
