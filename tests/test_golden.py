"""Committed golden vectors (tests/golden/bundled_pair.json, made by tests/golden/make_golden.py): the oracle must keep
reproducing them (CPU), and the HIP engine must match them through the C ABI (GPU)."""
import json
import os

import numpy as np
import pytest

from tests import util

GOLD = json.load(open(os.path.join(util.ROOT, "tests", "golden", "bundled_pair.json")))
SEARCH = {"vgicp_direct1": 2, "vgicp_direct7": 1, "vgicp_direct27": 0}


@pytest.mark.parametrize("case", sorted(SEARCH))
def test_oracle_reproduces_golden(case):
    from oracle import oracle as O
    tgt, src = util.bundled_pair()
    assert [len(tgt), len(src)] == GOLD["counts_head_preprocessing"]
    g = O.FastVGICP(search=SEARCH[case])
    g.set_target(tgt); g.set_source(src)
    r = g.align()
    ref = GOLD["cases"][case]
    assert util.rel_err(r["T"], ref["T"]) < 1e-9 and r["iterations"] == ref["iterations"]
    assert abs(g.fitness() - ref["fitness"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(SEARCH))
def test_engine_matches_golden_vgicp(case):
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    ref = GOLD["cases"][case]
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(SEARCH[case])
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    for pn, T in (("identity", np.eye(4)), ("relative_txt", util.relative_pose())):
        e, H, b = c.linearize(T)
        lin = ref["linearize"][pn]
        assert c.get_num_correspondences() == lin["num_correspondences"]
        assert abs(e - lin["error"]) <= 1e-5 * abs(lin["error"])   # fp32-stored covariances vs the all-fp64 oracle
        assert util.rel_err(H, lin["H"]) < 1e-5 and util.rel_err(b, lin["b"]) < 1e-5
    r = c.align()
    assert r["converged"] == ref["converged"] and r["num_linearize"] == ref["num_linearize"] and r["num_error_evals"] == ref["num_error_evals"]
    assert util.rel_err(r["T"], ref["T"]) < 1e-4 and util.rel_err(r["H"], ref["H"]) < 1e-4   # north_star tolerance
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert abs(f - ref["fitness"]) <= 1e-4 * ref["fitness"]
    assert len(c.get_voxelmap()[0]) == ref["num_voxels"]
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case,mode", [("ndt_d2d", 1), ("ndt_p2d", 0)])
def test_engine_matches_golden_ndt(case, mode):
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    ref = GOLD["cases"][case]
    c = capi.NDTCore(0)
    c.set_distance_mode(mode)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    r = c.align()
    assert r["converged"] == ref["converged"]
    assert util.rel_err(r["T"], ref["T"]) < 1e-4
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert abs(f - ref["fitness"]) <= 1e-4 * ref["fitness"]
    c.close()


@pytest.mark.parametrize("case", ["gicp", "gicp_maxdist1"])
def test_oracle_reproduces_golden_gicp(case):
    from oracle import oracle as O
    tgt, src = util.bundled_pair()
    ref = GOLD["cases"][case]
    g = O.FastVGICP()
    g.set_gicp_mode(True, 3.4028234663852886e38 if ref["max_correspondence_distance"] is None else ref["max_correspondence_distance"])
    g.set_target(tgt); g.set_source(src)
    r = g.align()
    assert util.rel_err(r["T"], ref["T"]) < 1e-9 and r["iterations"] == ref["iterations"]
    assert abs(g.fitness() - ref["fitness"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["gicp", "gicp_maxdist1"])
def test_engine_matches_golden_gicp(case):
    from fast_gicp_amd import capi, distributed as D
    tgt, src = util.bundled_pair()
    ref = GOLD["cases"][case]
    c = capi.VGICPCore(0)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3)
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    if ref["max_correspondence_distance"] is not None:
        c.gicp_set_max_correspondence_distance(ref["max_correspondence_distance"])
    for pn, T in (("identity", np.eye(4)), ("relative_txt", util.relative_pose())):
        e, H, b = c.gicp_linearize(T)
        lin = ref["linearize"][pn]
        assert int((c.gicp_get_correspondences() >= 0).sum()) == lin["num_correspondences"]
        assert abs(e - lin["error"]) <= 1e-5 * abs(lin["error"])   # fp32-stored covariances vs the all-fp64 oracle
        assert util.rel_err(H, lin["H"]) < 1e-5 and util.rel_err(b, lin["b"]) < 1e-5
    r = D.ShardedLsq(lambda T: c.gicp_linearize(T), lambda T: c.gicp_compute_error(T, derivatives=False), lambda v: v).align()
    assert r["converged"] == ref["converged"]
    assert util.rel_err(r["T"], ref["T"]) < 1e-4 and util.rel_err(r["H"], ref["H"]) < 1e-4   # north_star tolerance
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert abs(f - ref["fitness"]) <= 1e-4 * ref["fitness"]
    c.close()
