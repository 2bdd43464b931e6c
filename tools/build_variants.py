#!/usr/bin/env python
"""A/B builds of libfast_vgicp_hip.so (same ABI, different compile-time choices) under fast_gicp_amd/lib/variants/<name>/,
selected at run time with FVH_LIB_PATH (fast_gicp_amd/capi.py). The .so files are git-ignored but travel with gpurun.
    python tools/build_variants.py name="flags" [name="flags" ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fast_gicp_amd", "csrc", "fvh_capi.hip")


def build(name, flags):
    out_dir = os.path.join(ROOT, "fast_gicp_amd", "lib", "variants", name)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libfast_vgicp_hip.so")
    sys.path.insert(0, ROOT)
    from fast_gicp_amd import build as _build
    cmd = ["/opt/rocm/bin/hipcc"] + _build.HIPCC_FLAGS + ["-o", out, SRC, "-ldl"] + flags.split()
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    procs = []
    for arg in sys.argv[1:]:
        name, _, flags = arg.partition("=")
        print(name, "->", build(name, flags), flush=True)
