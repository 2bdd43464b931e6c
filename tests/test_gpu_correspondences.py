"""Index-level parity of the correspondence search (find_voxel_correspondences.cu:16-111, ndt_cuda.cu:142-161,
fast_gicp_impl.hpp:118-156): the engine's (source element, target voxel) PAIRS -- fvh_vgicp_get_voxel_correspondences ([VC]:65),
fvh_ndt_get_voxel_correspondences, fvh_vgicp_gicp_get_correspondences -- against the oracle's lists. Exact equality: a count (or the
sums downstream) could hide two compensating index errors, a pair list cannot.

Voxels are compared by COORDINATE (nobody's numbering), source elements by point index (VGICP, NDT P2D) or by the coordinate of the
source voxel (NDT D2D). The fp64 oracle emits its list thread by thread, so it is compared as a set; the cuda-compat leg emits it
offset-major like the reference's device code and like the engine's getter: compared as a LIST. The 100k / 1M cases live in
tests/test_gpu_parity_large.py (they share its fixtures)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

D27, D7, D1, RADIUS = 0, 1, 2, 3


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def pair():
    return util.bundled_pair()


@pytest.fixture(scope="module")
def oracle_covs(O, pair):
    tgt, src = pair
    return O.covariances_knn(tgt, 20, O.PLANE), O.covariances_knn(src, 20, O.PLANE)


def _poses():
    return [np.eye(4), util.relative_pose(), util.random_pose(np.random.default_rng(7)), util.random_pose(np.random.default_rng(8), 3.0, 1.0) @ util.relative_pose()]


@pytest.mark.parametrize("search,radius,name", [(D1, 0.0, "DIRECT1"), (D7, 0.0, "DIRECT7"), (D27, 0.0, "DIRECT27"), (RADIUS, 1.5, "RADIUS1.5"), (RADIUS, 2.0, "RADIUS2")])
def test_vgicp_pairs_equal_oracle_17k(O, pair, oracle_covs, search, radius, name):
    from fast_gicp_amd import capi
    tgt, src = pair
    cov_t, cov_s = oracle_covs
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(search, radius)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.set_target_covariances(cov_t); c.set_source_covariances(cov_s)
    c.create_target_voxelmap()
    # (the CPU class has no DIRECT_RADIUS, fast_vgicp_voxel.hpp:10-43: for it the pairs are rebuilt below from the oracle's voxel set)
    g = O.FastVGICP(search=search if search != RADIUS else D27)
    g.set_target(tgt); g.set_source(src); g.set_target_covs(cov_t); g.set_source_covs(cov_s); g.prepare()
    for T in _poses():
        c.update_correspondences(T)
        if search != RADIUS:
            g.linearize(T)
            util.assert_same_correspondences(c, g)
        else:
            # DIRECT_RADIUS (fast_vgicp_cuda.cu:77-91): offsets of the ceil(r) cube with |o| <= r + 1e-3, probed around floor(q / res - 0.5)
            offs = np.array(O.neighbor_offsets(O.DIRECT_RADIUS, radius), np.int64).reshape(-1, 3)
            vox = set(map(tuple, np.asarray(O.voxelmap_vgicp(tgt, cov_t, 1.0)[0], np.int64)))
            q = src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
            base = np.floor(q / 1.0 - 0.5).astype(np.int64)
            ref = [(i, 0, 0, 0) + tuple(base[i] + o) for o in offs for i in range(len(src)) if tuple(base[i] + o) in vox]
            got = util.engine_corr_rows(c)
            assert len(got) == len(ref)
            assert np.array_equal(got, np.asarray(ref, np.int64)), "DIRECT_RADIUS pairs differ (offset-major order included)"
    c.close()


def test_vgicp_pairs_after_align_are_those_of_the_last_linearisation(O, pair):
    """align() leaves the correspondences of the last CONSUMED linearisation behind (what compute_error() then evaluates, [VC]:69-71);
    the oracle's list after its align() is the one of its last linearize()."""
    from fast_gicp_amd import capi
    tgt, src = pair
    for search in (D1, D27):
        c = capi.VGICPCore(0)
        c.set_neighbor_search_method(search)
        c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
        c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
        r = c.align()
        g = O.FastVGICP(search=search)
        g.set_target(tgt); g.set_source(src)
        g.set_target_covs(c.get_covariances("target").astype(np.float64)); g.set_source_covs(c.get_covariances("source").astype(np.float64))
        ro = g.align()
        assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
        util.assert_same_correspondences(c, g)
        c.close()


@pytest.mark.parametrize("search", [D1, D7, D27])
def test_vgicp_fp32_mode_list_equals_cuda_compat_in_order(O, pair, search):
    """FVH_COMPUTE_FP32 computes the voxel coordinate in float like vector3_hash.cuh:35-38: its list must equal the cuda-compat leg's
    (offset-major, find_voxel_correspondences.cu:84-111) element by element."""
    from fast_gicp_amd import capi
    tgt, src = pair
    g = O.CudaCompatVGICP(search=search)
    g.set_target(tgt); g.set_source(src); g.prepare()
    g.linearize(np.eye(4))  # (builds covariances and the voxel map)
    c = capi.VGICPCore(0)
    c.set_precision(capi.COMPUTE_FP32)
    c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.set_target_covariances(g.get_covs("target")); c.set_source_covariances(g.get_covs("source"))
    c.create_target_voxelmap()
    assert set(map(tuple, c.get_voxelmap()[0])) == set(map(tuple, g.get_voxelmap()[0]))
    for T in _poses():
        c.update_correspondences(T)
        g.linearize(T)
        util.assert_same_correspondences(c, g, ordered=True)
    c.close()


@pytest.fixture(scope="module")
def gicp_test_pair():
    return util.bundled_pair(origin_filter=False, leaf=0.2, exact_voxelgrid=True)


@pytest.mark.parametrize("mode,search", [(1, D7), (0, D7), (1, D1), (1, D27), (0, D27)])
def test_ndt_pairs_equal_oracle(O, gicp_test_pair, mode, search):
    from fast_gicp_amd import capi
    tgt, src = gicp_test_pair
    c = capi.NDTCore(0)
    c.set_distance_mode(mode); c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.create_voxelmaps()
    g = O.NDT(mode=mode, search=search)
    g.set_target(tgt); g.set_source(src); g.prepare()
    for T in _poses():
        c.update_correspondences(T)
        g.linearize(T)
        util.assert_same_correspondences(c, g, d2d=(mode == 1))
    # ... and after an align
    r = c.align()
    ro = g.align()
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
    util.assert_same_correspondences(c, g, d2d=(mode == 1))
    c.close()


@pytest.mark.parametrize("max_dist", [None, 1.0, 0.3])
def test_gicp_nearest_point_list_equals_oracle(O, pair, max_dist):
    """fvh_vgicp_gicp_get_correspondences against the oracle's FastGICP list (fast_gicp_impl.hpp:118-156): (source point, target point)."""
    from fast_gicp_amd import capi
    tgt, src = pair
    c = capi.VGICPCore(0)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
    if max_dist is not None:
        c.gicp_set_max_correspondence_distance(max_dist)
    g = O.FastVGICP(k=20)
    g.set_gicp_mode(True, 3.4028234663852886e38 if max_dist is None else max_dist)
    g.set_target(tgt); g.set_source(src)
    g.set_target_covs(c.get_covariances("target").astype(np.float64)); g.set_source_covs(c.get_covariances("source").astype(np.float64))
    for T in _poses():
        c.gicp_update_correspondences(T)
        g.linearize(T)
        corr = c.gicp_get_correspondences()
        got = np.stack([np.nonzero(corr >= 0)[0], corr[corr >= 0]], axis=1)
        ref = g.correspondences()[:, [0, 4]]
        assert np.array_equal(got, util.sort_rows(ref))
    c.close()


def test_stored_correspondences_survive_a_spatial_order_that_appears_later():
    """Round 6: on clouds walked in Morton order the correspondence rows are indexed by the element's POSITION in that walk (Engine::corr_by_position).
    The indexing is decided by the launch that FINDS a list and kept for every later evaluation of it. Here the list is found while the source has no
    spatial order (rows by point index); then an order appears (fitness_score sorts the source, and the handle's coherent_min_points is lowered so that
    the cost kernel now walks in it): compute_error must still read the rows the way they were written -- same error, H, b as before -- and the getters
    still return the same pairs; a list found WITH the order (rows by position) gives the same pairs and sums again."""
    from fast_gicp_amd import capi
    tgt, src, T = util.synthetic_pair(20000, 6000, seed=5, extent=30.0)
    c = capi.VGICPCore(0)
    c.set_resolution(0.5); c.set_neighbor_search_method(capi.DIRECT7)
    c.set_engine_params(coherent_min_points=1 << 30)  # nothing walks in Morton order yet
    c.set_target_cloud(tgt); c.set_target_covariances(np.tile(np.eye(3) * 0.01, (len(tgt), 1, 1))); c.create_target_voxelmap()
    c.set_source_cloud(src); c.set_source_covariances(np.tile(np.eye(3) * 0.01, (len(src), 1, 1)))
    c.update_correspondences(T)                       # found without an order: rows by point index
    e0, H0, b0 = c.compute_error(T)
    pairs0 = c.get_voxel_correspondences().copy()
    assert len(pairs0) > 1000
    c.set_engine_params(coherent_min_points=1000)
    c.fitness_score(T)                                # sorts the source: from here on the cost kernel walks it in Morton order
    e1, H1, b1 = c.compute_error(T)
    assert abs(e1 - e0) <= 1e-12 * abs(e0) and util.rel_err(H1, H0) < 1e-12 and util.rel_err(b1, b0) < 1e-11  # (another summation order)
    assert np.array_equal(c.get_voxel_correspondences(), pairs0)
    c.update_correspondences(T)                       # found again, now WITH the order: rows by position
    e2, H2, b2 = c.compute_error(T)
    assert abs(e2 - e0) <= 1e-12 * abs(e0) and util.rel_err(H2, H0) < 1e-12
    assert np.array_equal(c.get_voxel_correspondences(), pairs0)
    r = c.align(T)
    assert r["converged"]
    assert len(c.get_voxel_correspondences()) == c.get_num_correspondences() > 1000
    c.close()
