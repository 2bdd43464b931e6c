"""swap_source_and_target() right after an align builds the new target map on the engine's side stream, beside the caller's
next source chain (fvh_capi.hip: Engine). The map, the poses and every getter must be what the one-stream order gives, and
any call that is not part of the source chain must wait for the build."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

LOOP = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from tests import util
from fast_gicp_amd import capi
rng = np.random.default_rng(5)
base = rng.uniform(-15, 15, size=(9000, 3)).astype(np.float32); base[:, 2] *= 0.15
frames = []
for i in range(7):
    T = util.random_pose(np.random.default_rng(100 + i), 0.15, 0.03)
    frames.append(((base[rng.permutation(9000)[:8000]] @ T[:3, :3].T + T[:3, 3]) + rng.normal(0, 0.01, size=(8000, 3))).astype(np.float32))
c = capi.VGICPCore(0)
c.set_neighbor_search_method(1)
c.set_target_cloud(frames[0]); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
c.set_source_cloud(frames[1]); c.find_source_neighbors(20); c.calculate_source_covariances(3)
out = [c.align()["T"]]
nvox = []
for f in frames[2:]:
    c.swap_source_and_target()
    c.set_source_cloud(f); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    r = c.align()
    out.append(r["T"]); nvox.append(len(c.get_voxelmap()[0]))
np.savez(sys.argv[1], T=np.stack(out), nvox=np.array(nvox))
"""


def test_the_reuse_loop_gives_the_same_poses_with_and_without_the_side_stream(tmp_path):
    res = []
    for on in ("0", "1"):
        path = str(tmp_path / ("side%s.npz" % on))
        subprocess.check_call([sys.executable, "-c", LOOP % util.ROOT, path], env=dict(os.environ, FVH_SIDE_STREAM=on), cwd=util.ROOT)
        res.append(np.load(path))
    assert np.array_equal(res[0]["nvox"], res[1]["nvox"])
    # (two builds of one map differ in the last bits: fp64 atomics accumulate in arrival order)
    assert util.rel_err(res[0]["T"], res[1]["T"]) < 1e-9


def test_calls_outside_the_source_chain_wait_for_the_side_build():
    from fast_gicp_amd import capi
    rng = np.random.default_rng(2)
    a = rng.uniform(-10, 10, size=(6000, 3)).astype(np.float32); a[:, 2] *= 0.2
    b = (a[:5000] + np.array([0.1, 0.05, 0.0], np.float32)).astype(np.float32)
    d = rng.uniform(-10, 10, size=(4000, 3)).astype(np.float32)
    ref = capi.VGICPCore(0); ref.set_neighbor_search_method(1)
    ref.set_target_cloud(b); ref.find_target_neighbors(20); ref.calculate_target_covariances(3); ref.create_target_voxelmap()
    want = ref.get_voxelmap()
    want_voxels = len(want[0])
    key = lambda v: sorted(map(tuple, v[0].tolist()))

    c = capi.VGICPCore(0); c.set_neighbor_search_method(1)
    c.set_target_cloud(a); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(b); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    for trial in range(20):
        c.align()
        c.swap_source_and_target()           # b becomes the target: its map is built on the side stream
        if trial % 4 == 0:                   # a getter right behind the swap reads the finished map
            assert len(c.get_voxelmap()[0]) == want_voxels
        elif trial % 4 == 1:                 # the full map
            got = c.get_voxelmap()
            assert key(got) == key(want) and int(got[1].sum()) == int(want[1].sum())
        elif trial % 4 == 2:                 # replacing the target while its old map is being built, then rebuilding
            c.set_target_cloud(d); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
            c.set_target_cloud(b); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
            assert len(c.get_voxelmap()[0]) == want_voxels
        c.set_source_cloud(a); c.find_source_neighbors(20); c.calculate_source_covariances(3)
        r = c.align()
        assert r["converged"]
        c.swap_source_and_target()           # back: a is the target again
        c.set_source_cloud(b); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    c.close(); ref.close()
