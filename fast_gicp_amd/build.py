"""Builds libfast_vgicp_hip.so (the C-ABI HIP engine) in-tree for gfx950."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libfast_vgicp_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("fvh_capi.hip", "host_runtime.inc.hpp", "host_stages.inc.hpp", "host_gicp.inc.hpp", "host_downsample.inc.hpp", "kernels_compat.hpp", "kernels_cost.hpp", "kernels_cov.hpp", "kernels_voxelmap.hpp", "kernels_sort.hpp", "kernels_downsample.hpp", "kernels_peer.hpp", "dev_math.hpp")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "fast_vgicp_hip.h")
# -disable-machine-licm: the persistent LM kernel is one big loop over LM transitions; machine LICM hoists ~25 constant
# materialisations and thread-index-derived addresses of its epilogue into the kernel prologue, where they stay live across
# the main loop (187 instead of 162 VGPRs = 2 instead of 3 workgroups per CU). tools/kernel_resources.py reports the effect.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-mllvm", "-disable-machine-licm"]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in SOURCES + [HEADER])


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> fast_gicp_amd/lib/libfast_vgicp_hip.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH, SOURCES[0], "-ldl"] + os.environ.get("FVH_EXTRA_HIPCC_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


TEST_LIB_PATH = os.path.join(LIB_DIR, "variants", "test_kernels", "libfast_vgicp_hip.so")


def build_test_kernels_lib(force=False, verbose=False):
    """The same library with -DFVH_TEST_KERNELS: the superseded full-sweep / eight-queries-per-wave kernels and the FVH_*_MODE
    switches that select them -- cross-checks of the culled kernels, used by tests/test_gpu_edge_and_properties.py only
    (FVH_LIB_PATH points a test subprocess at it). The product library does not contain them."""
    if not force and os.path.exists(TEST_LIB_PATH):
        t = os.path.getmtime(TEST_LIB_PATH)
        if not any(os.path.exists(s) and os.path.getmtime(s) > t for s in SOURCES + [HEADER]):
            return TEST_LIB_PATH
    os.makedirs(os.path.dirname(TEST_LIB_PATH), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-DFVH_TEST_KERNELS", "-o", TEST_LIB_PATH, SOURCES[0], "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return TEST_LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_test_kernels_lib(force=True, verbose=True))
