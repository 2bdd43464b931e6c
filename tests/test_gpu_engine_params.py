"""fvh_engine_params (round 6): routes / thresholds / watchdogs are per-HANDLE state -- two handles of one process take different
routes, results bit for bit the same; bad values are refused."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _prepared(c, tgt, src):
    from fast_gicp_amd import capi
    c.set_neighbor_search_method(capi.DIRECT7)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()


def test_two_handles_take_different_routes_and_agree():
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    a, b = capi.VGICPCore(0), capi.VGICPCore(0)
    d = capi.default_engine_params()
    assert a.get_engine_params().persistent == d.persistent == 1 and a.get_engine_params().sort_mode == d.sort_mode
    b.set_engine_params(persistent=0, sort_mode=1, cost_group_max=4, zerocopy_result=0, side_stream=0)
    assert b.get_engine_params().persistent == 0 and a.get_engine_params().persistent == 1  # one handle only
    r0 = np.array(capi.debug_sort_routes())
    _prepared(a, tgt, src)
    ra = a.align()
    r1 = np.array(capi.debug_sort_routes())
    _prepared(b, tgt, src)
    rb = b.align()
    r2 = np.array(capi.debug_sort_routes())
    assert tuple(r1 - r0)[:2] == (2, 0) and tuple(r2 - r1)[:2] == (0, 2)  # a: cooperative sort; b: the one-workgroup sort
    assert ra["num_launches"] == 1 and rb["num_launches"] > 1                # a: persistent LM kernel; b: one launch per transition
    assert np.array_equal(ra["T"], rb["T"]) and np.array_equal(ra["H"], rb["H"]) and ra["num_error_evals"] == rb["num_error_evals"]
    # the watchdog of ONE handle at zero: its persistent launch aborts and is redone per transition -- same bits
    a.set_engine_params(persist_watchdog_ticks=0)
    ra2 = a.align()
    assert a.debug_persist_aborts() == 1 and ra2["num_launches"] > 1 and np.array_equal(ra2["T"], ra["T"])
    a.close(); b.close()


def test_bad_engine_params_are_refused():
    from fast_gicp_amd import capi
    c = capi.VGICPCore(0)
    for bad in (dict(sort_mode=7), dict(knn_block=100), dict(cost_group_max=0), dict(cost_group_max=9), dict(cost_max_blocks=5000), dict(sort_fused_bits=8), dict(lm_everywhere=3)):
        with pytest.raises(capi.FvhError):
            c.set_engine_params(**bad)
    with pytest.raises(capi.FvhError):
        c.set_engine_params(no_such_field=1)
    p = c.get_engine_params()
    p.struct_size = 4  # a caller built against another header
    with pytest.raises(capi.FvhError):
        c._call("set_engine_params", __import__("ctypes").byref(p))
    assert c.get_engine_params().sort_mode == capi.default_engine_params().sort_mode  # nothing stuck
    n = capi.NDTCore(0)
    n.set_engine_params(persistent=0)
    assert n.get_engine_params().persistent == 0
    n.close(); c.close()
