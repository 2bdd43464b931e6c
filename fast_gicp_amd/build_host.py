"""Builds the host-side native pieces (g++): the pygicp pybind11 module and the gicp_align CLI."""
import os
import subprocess
import sys
import sysconfig

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
INCLUDE = os.path.join(ROOT, "include")
PYGICP_SRC = os.path.join(_HERE, "python", "pygicp.cpp")
PYGICP_SO = os.path.join(_HERE, "pygicp" + sysconfig.get_config_var("EXT_SUFFIX"))
ALIGN_SRC = os.path.join(_HERE, "apps", "gicp_align.cpp")
ALIGN_BIN = os.path.join(_HERE, "apps", "gicp_align")
HEADERS = [os.path.join(INCLUDE, "fast_gicp_amd", f) for f in ("registration.hpp", "kdtree.hpp", "voxelgrid.hpp", "pcd_io.hpp")] + [_build.HEADER]


def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build_pygicp(force=False):
    _build.build_lib()
    if not force and not _stale(PYGICP_SO, [PYGICP_SRC] + HEADERS):
        return PYGICP_SO
    import pybind11
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fopenmp", "-ffp-contract=off", "-I", INCLUDE, "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
           PYGICP_SRC, "-o", PYGICP_SO, "-L", _build.LIB_DIR, "-lfast_vgicp_hip", "-Wl,-rpath,$ORIGIN/lib"]
    subprocess.check_call(cmd)
    return PYGICP_SO


def build_align(force=False):
    if not os.path.exists(ALIGN_SRC):
        return None
    _build.build_lib()
    if not force and not _stale(ALIGN_BIN, [ALIGN_SRC] + HEADERS):
        return ALIGN_BIN
    cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I", INCLUDE, ALIGN_SRC, "-o", ALIGN_BIN, "-L", _build.LIB_DIR, "-lfast_vgicp_hip", "-Wl,-rpath,$ORIGIN/../lib"]
    subprocess.check_call(cmd)
    return ALIGN_BIN


KITTI_SRC = os.path.join(os.path.dirname(ALIGN_SRC), "gicp_kitti.cpp")
KITTI_BIN = os.path.join(os.path.dirname(ALIGN_BIN), "gicp_kitti")


def build_kitti(force=False):
    if not os.path.exists(KITTI_SRC):
        return None
    _build.build_lib()
    if not force and not _stale(KITTI_BIN, [KITTI_SRC] + HEADERS):
        return KITTI_BIN
    cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I", INCLUDE, KITTI_SRC, "-o", KITTI_BIN, "-L", _build.LIB_DIR, "-lfast_vgicp_hip", "-Wl,-rpath,$ORIGIN/../lib"]
    subprocess.check_call(cmd)
    return KITTI_BIN


def build_all(force=False):
    return [build_pygicp(force), build_align(force), build_kitti(force)]


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
