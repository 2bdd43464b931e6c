"""`pip install .` / `python setup.py build_ext --inplace` for the MI355X engine: builds libfast_vgicp_hip.so (hipcc, gfx950) and the
pygicp module on top of its C ABI -- the role of koide3/fast_gicp's setup.py (a CMake-driven build_ext there; here the two compiler
invocations of fast_gicp_amd/build.py and build_host.py, the same ones __graft_entry__.build() runs, so nothing needs CMake).
The shared objects stay IN-TREE (fast_gicp_amd/lib/, fast_gicp_amd/pygicp*.so): `import pygicp` resolves through pygicp.py."""
import os
import sys

from setuptools import Extension, setup
from setuptools.command.build_ext import build_ext

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class HipBuild(build_ext):
    def run(self):
        from fast_gicp_amd import build, build_host
        print("hipcc ->", build.build_lib(force=self.force, verbose=True))
        for out in build_host.build_all(force=self.force):
            print("g++   ->", out)

    def get_outputs(self):
        return []


setup(
    name="pygicp",
    version="0.0.3",  # the reference's (koide3/fast_gicp package.xml / setup.py)
    description="pygicp (fast_gicp Python bindings) on the MI355X HIP engine",
    packages=["fast_gicp_amd"],
    py_modules=["pygicp"],
    package_data={"fast_gicp_amd": ["lib/*.so", "pygicp*.so", "csrc/*", "python/*", "apps/*.cpp"]},
    ext_modules=[Extension("fast_gicp_amd.pygicp", sources=[])],  # placeholder: HipBuild does the work
    cmdclass={"build_ext": HipBuild},
    zip_safe=False,
)
