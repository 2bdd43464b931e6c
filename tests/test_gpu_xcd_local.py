"""The persistent LM kernel's hand-off flavours and grid layouts (kernels_cost.hpp: xcd_local, ng): whatever travels
through an XCD's own L2 instead of write-through + memory-side polls, and however the grid reduces -- chip-wide in eight groups,
chip-wide in one group -- an align must give the SAME BITS as the per-transition route on the same layout,
the placement check must never fire on this hardware, and XCD-local vs write-through hand-offs must not change a single bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

RUN = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from tests import util
from fast_gicp_amd import capi, workloads
tgt, src = util.bundled_pair()
out = {}
for search in (0, 2):  # DIRECT27: 474 workgroups, eight groups; DIRECT1: 68 workgroups, a "small grid"
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    rs = [c.align() for _ in range(3)]
    assert all(r["num_launches"] == 1 for r in rs) and c.debug_persist_aborts() == 0, (rs, c.debug_persist_aborts())
    assert all(np.array_equal(r["T"], rs[0]["T"]) and np.array_equal(r["H"], rs[0]["H"]) for r in rs)
    out["vgicp%%d_T" %% search], out["vgicp%%d_H" %% search], out["vgicp%%d_grid" %% search] = rs[0]["T"], rs[0]["H"], np.array(c.debug_persist_grid())
    c.close()
f0, f1 = workloads.lidar_frame(3), workloads.lidar_frame(4)
vg = capi.VoxelGrid(0)
for mode in (1, 0):
    d = capi.NDTCore(0)
    d.set_distance_mode(mode); d.set_neighbor_search_method(1); d.set_resolution(1.0)
    d.set_target_cloud(vg.filter(f0, 0.25, vg.APPROXIMATE)); d.set_source_cloud(vg.filter(f1, 0.25, vg.APPROXIMATE))
    d.align()  # (D2D: the second align's grid is shaped by the source-voxel count the first one saw)
    rs = [d.align() for _ in range(3)]
    assert all(r["num_launches"] == 1 for r in rs)
    assert all(np.array_equal(r["T"], rs[0]["T"]) for r in rs)
    out["ndt%%d_T" %% mode], out["ndt%%d_H" %% mode] = rs[0]["T"], rs[0]["H"]
    d.close()
out["xcd_local"] = np.array(capi.debug_xcd_local())
np.savez(sys.argv[1], **out)
"""


def _run(tmp_path, name, **env):
    path = str(tmp_path / (name + ".npz"))
    subprocess.check_call([sys.executable, "-c", RUN % util.ROOT, path], env=dict(os.environ, **env), cwd=util.ROOT)
    return np.load(path)


@pytest.mark.parametrize("layout", ["0", "2"])
def test_local_and_write_through_handoffs_give_the_same_bits(tmp_path, layout):
    """Same layout (FVH_SMALL_GRID_LAYOUT), hand-offs through the XCD's L2 (default) vs write-through everywhere (FVH_XCD_LOCAL=0) vs
    one launch per LM transition (FVH_PERSISTENT=0): the sums are added in the same order on all three -> identical poses and
    Hessians, for VGICP (large and small grid) and NDT (P2D, D2D). The placement check must not have ended a single launch."""
    a = _run(tmp_path, "local", FVH_SMALL_GRID_LAYOUT=layout)
    b = _run(tmp_path, "wt", FVH_SMALL_GRID_LAYOUT=layout, FVH_XCD_LOCAL="0")
    assert tuple(a["xcd_local"]) == (1, 0), a["xcd_local"]   # still wanted, no placement abort
    assert tuple(b["xcd_local"]) == (0, 0)
    for k in a.files:
        if k.startswith("ndt1"):  # NDT D2D walks the source voxels in the order of the compact list, i.e. of workgroup arrival in the map build:
            assert util.rel_err(a[k], b[k]) < 1e-12, (layout, k)  # two PROCESSES (two builds) agree to rounding, not to the bit
        elif k != "xcd_local":
            assert np.array_equal(a[k], b[k]), (layout, k)
    if layout == "2":  # the per-transition route: no persistent launch at all (the launch counts asserted in RUN do not apply -> its own script would be needed);
        return         # it is compared with the persistent route, bit for bit, in tests/test_gpu_parity.py and tests/test_gpu_edge_and_properties.py


def test_layouts_agree_to_rounding(tmp_path):
    """Different layouts partition and order the sums differently: not the same bits, but the same registration (1e-9)."""
    runs = [_run(tmp_path, "l" + l, FVH_SMALL_GRID_LAYOUT=l) for l in ("0", "2")]
    for r in runs[1:]:
        for k in runs[0].files:
            if k.endswith("_T"):
                assert util.rel_err(r[k], runs[0][k]) < 1e-9, k
    # the large grid (DIRECT27 at 17k points: eight groups in every layout) is the same launch in all three
    assert all(np.array_equal(r["vgicp0_T"], runs[0]["vgicp0_T"]) for r in runs[1:])


def test_lm_step_on_every_workgroup_gives_the_collectors_bits(tmp_path):
    """CostParams::lm_everywhere (default on grids of <= 2 workgroups per CU): every workgroup polls the group rows and runs the LM step on its
    own copy of the state instead of waiting for a collector's broadcast. Same sums, same instructions: identical poses and Hessians to the
    collectors-only protocol (FVH_LM_EVERYWHERE=0) and to the always-on flavour (=2), for VGICP (474 and 68 workgroups) and NDT P2D; NDT D2D
    to rounding (two processes, two map builds: see above)."""
    a = _run(tmp_path, "rule")
    for name, env in (("never", dict(FVH_LM_EVERYWHERE="0")), ("always", dict(FVH_LM_EVERYWHERE="2"))):
        b = _run(tmp_path, name, **env)
        assert tuple(b["xcd_local"])[1] == 0
        for k in a.files:
            if k == "xcd_local" or k.endswith("_grid"):
                continue
            if k.startswith("ndt1"):
                assert util.rel_err(a[k], b[k]) < 1e-12, (name, k)
            else:
                assert np.array_equal(a[k], b[k]), (name, k)
