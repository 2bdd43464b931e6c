"""GPU parity tests at the sizes of BASELINE.json configs[2], [3] and [4] -- the HIP engine through the C ABI against the
ORACLE on the same inputs (not properties): the large-cloud code paths (3-pass radix sort, Morton-permuted cost walk,
4-offset work items at 100k points, dense key array of a 1M-point map) meet the oracle here.

  C3  synthetic 100k <-> 100k, seed 42, res 0.5, RBF 0.5/2.5 and k-NN covariances, DIRECT1 + DIRECT27
  C5  synthetic 1M-point map <-> 100k-point scan, seed 44, res 0.5, DIRECT7
  C4  two consecutive ~118k-point simulated 64-ring LiDAR frames -> ApproximateVoxelGrid 0.25 -> NDT D2D, DIRECT7, res 1.0
      (+ the raw 118k-point frame through NDT P2D)

Tolerances, written here on purpose:
  * k-NN index lists, voxel coordinate sets, per-voxel point counts, correspondence counts: EXACT;
  * voxel means / covariances: fp32 storage rounding;
  * covariances after an eigen-based regularisation: max error over ALL points against a conditioning-aware bound
    (util.cov_error_bound: storage rounding + input perturbation x lambda_max / eigen-gap) -- no quantiles;
  * err / H / b at three poses: rel 1e-9 against the oracle fed the SAME fp32-stored covariances / voxel records
    (compute_derivatives.cu:50-103, ndt_compute_derivatives.cu:104-175), rel 1e-5 (VGICP) / 2e-5 (NDT) against all-fp64;
    the gradient b is measured on its Cauchy-Schwarz scale sqrt(H_ii * err) (util.sums_close), H and err on their own;
  * final transform, final Hessian, fitness: 1e-4 relative (north_star) with EQUAL linearisation / error-evaluation counts.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

D27, D7, D1 = 0, 1, 2
PLANE = 3


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _core():
    from fast_gicp_amd import capi
    return capi.VGICPCore(0)


def _poses(T_gt, seed):
    return [np.eye(4), T_gt, util.random_pose(np.random.default_rng(seed), 1.0, 0.3) @ T_gt]


def _check_sums(c, refs_tols, poses, seed, d2d=False):
    """linearize + trial-step error of the engine `c` against oracle objects [(oracle, tol), ...] at each pose; the correspondence
    PAIRS (source element, voxel coordinate) equal the first oracle's exactly (index-level parity, tests/test_gpu_correspondences.py)."""
    for T in poses:
        e, H, b = c.linearize(T)
        n_corr = c.get_num_correspondences()
        refs_tols[0][0].linearize(T)
        util.assert_same_correspondences(c, refs_tols[0][0], d2d=d2d)
        T2 = util.random_pose(np.random.default_rng(seed), 0.2, 0.05) @ T
        e2 = c.compute_error(T2, derivatives=False)
        for g, tol in refs_tols:
            eo, Ho, bo = g.linearize(T)
            assert n_corr == g.num_correspondences()
            assert util.sums_close(e, H, b, eo, Ho, bo, tol), (tol, e, eo, util.rel_err(H, Ho), util.rel_err(b, bo))
            e2o = g.compute_error(T2)
            assert abs(e2 - e2o) <= tol * abs(e2o)


# =====================================================================================================
# C3: 100k <-> 100k, res 0.5
# =====================================================================================================
@pytest.fixture(scope="module")
def c3():
    return util.synthetic_pair(100_000, 100_000, seed=42)


@pytest.fixture(scope="module")
def c3_knn(O, c3):
    tgt, src, _ = c3
    return O.knn(tgt, 20), O.knn(src, 20)


def test_c3_knn_lists_exact(c3, c3_knn):
    tgt, src, _ = c3
    c = _core()
    c.set_target_cloud(tgt); c.find_target_neighbors(20)
    c.set_source_cloud(src); c.find_source_neighbors(20)
    assert np.array_equal(c.get_neighbors("target"), c3_knn[0])
    assert np.array_equal(c.get_neighbors("source"), c3_knn[1])
    c.close()


@pytest.mark.parametrize("reg", [3, 1])
def test_c3_knn_covariances(O, c3, c3_knn, reg):
    _, src, _ = c3
    c = _core()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(reg)
    got = c.get_covariances("source").astype(np.float64)
    ref = O.covariances_knn(src, 20, reg, idx=c3_knn[1])
    raw = O.covariances_knn(src, 20, O.NONE, idx=c3_knn[1])
    err, bound, degenerate = util.cov_error_bound(got, ref, raw, input_rel=1e-13, gaps="01" if reg == PLANE else "min")
    assert degenerate.mean() < 1e-3
    assert np.all(err[~degenerate] <= bound[~degenerate]), float((err / bound)[~degenerate].max())
    c.close()


def test_c3_rbf_covariances(O, c3):
    _, src, _ = c3
    c = _core()
    c.set_kernel_params(0.5, 2.5)
    c.set_source_cloud(src)
    c.calculate_source_covariances_rbf(0)
    got_raw = c.get_covariances("source").astype(np.float64)
    raw = O.covariances_rbf(src, 0.5, 2.5, O.NONE)
    scale = np.abs(raw).max(axis=(1, 2))  # (0 for an isolated clutter point: the only neighbour within 2.5 m is itself)
    err_raw = np.abs(got_raw - raw).max(axis=(1, 2))
    assert np.all(err_raw <= 5e-5 * scale + 1e-12), float((err_raw / np.maximum(scale, 1e-12)).max())  # fp32 weighted sums over up to thousands of neighbours (covariance_estimation_rbf.cu:40-109 is fp32 too)
    c.calculate_source_covariances_rbf(PLANE)
    got = c.get_covariances("source").astype(np.float64)
    ref = O.covariances_rbf(src, 0.5, 2.5, O.PLANE)
    err, bound, degenerate = util.cov_error_bound(got, ref, raw, input_rel=5e-5)
    # degenerate here = clutter points with 0 or 1 other point within 2.5 m (zero / rank-1 covariance: no plane normal exists);
    # ~1 % of this scene. Every other point is held to its bound.
    assert degenerate.mean() < 2e-2
    assert np.all(err[~degenerate] <= bound[~degenerate]), float((err / bound)[~degenerate].max())
    c.close()


@pytest.fixture(scope="module")
def c3_engine(c3):
    """Engine prepared as the bench does it (k-NN covariances through the Morton-sorted path), + its fp32-stored covariances."""
    tgt, src, _ = c3
    c = _core()
    c.set_resolution(0.5)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(PLANE); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(PLANE)
    yield c, c.get_covariances("target").astype(np.float64), c.get_covariances("source").astype(np.float64)
    c.close()


def test_c3_voxelmap_exact(O, c3, c3_engine):
    tgt, _, _ = c3
    c, cov_t, _ = c3_engine
    coords, num, means, covs = c.get_voxelmap()
    oc, on, om, ocv = O.voxelmap_vgicp(tgt, cov_t, 0.5)
    assert len(coords) == len(oc) == len(np.unique(coords, axis=0))
    order_g, order_o = np.lexsort(coords.T), np.lexsort(oc.T)
    assert np.array_equal(coords[order_g], oc[order_o]), "voxel coordinate sets differ"
    assert np.array_equal(num[order_g], on[order_o]) and int(num.sum()) == len(tgt)
    np.testing.assert_allclose(means[order_g], om[order_o], rtol=0, atol=4e-6 * np.abs(om).max())
    np.testing.assert_allclose(covs[order_g], ocv[order_o], rtol=0, atol=2e-7)


@pytest.mark.parametrize("search", [D1, D27])
def test_c3_linearize(O, c3, c3_engine, search):
    tgt, src, T = c3
    c, cov_t, cov_s = c3_engine
    c.set_neighbor_search_method(search)
    refs = []
    for rnd, tol in ((True, 1e-9), (False, 1e-5)):
        g = O.FastVGICP(search=search, resolution=0.5, round_fp32=rnd)
        g.set_target(tgt); g.set_source(src); g.set_target_covs(cov_t); g.set_source_covs(cov_s); g.prepare()
        refs.append((g, tol))
    _check_sums(c, refs, _poses(T, 7), 11)


@pytest.mark.parametrize("cov,search", [("rbf", D27), ("rbf", D1), ("knn", D27)])
def test_c3_align(O, c3, cov, search):
    """BASELINE configs[2]: the full registration, engine-estimated covariances against oracle-estimated ones."""
    tgt, src, T = c3
    c = _core()
    c.set_resolution(0.5); c.set_neighbor_search_method(search); c.set_kernel_params(0.5, 2.5)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    if cov == "rbf":
        c.calculate_target_covariances_rbf(PLANE); c.calculate_source_covariances_rbf(PLANE)
    else:
        c.find_target_neighbors(20); c.calculate_target_covariances(PLANE)
        c.find_source_neighbors(20); c.calculate_source_covariances(PLANE)
    c.create_target_voxelmap()
    r = c.align()
    g = O.FastVGICP(search=search, resolution=0.5, cov_mode=1 if cov == "rbf" else 0, kernel_width=0.5, kernel_max_dist=2.5)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4 and util.rel_err(r["H"], ro["H"]) < 1e-4
    f, fo = c.fitness_score(r["T"].astype(np.float32).astype(np.float64)), g.fitness()
    assert abs(f - fo) <= 1e-4 * fo
    te, re_ = util.pose_error(T, r["T"])
    assert te < 0.02 and re_ < np.radians(0.1)
    c.close()


# =====================================================================================================
# C5: 1M-point map <-> 100k-point scan, res 0.5, DIRECT7
# =====================================================================================================
@pytest.fixture(scope="module")
def c5():
    return util.synthetic_pair(1_000_000, 100_000, seed=44, extent=150.0)


@pytest.fixture(scope="module")
def c5_engine(c5):
    tgt, src, _ = c5
    c = _core()
    c.set_resolution(0.5); c.set_neighbor_search_method(D7)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(PLANE); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(PLANE)
    yield c
    c.close()


def test_c5_knn_lists_exact_1m(O, c5, c5_engine):
    tgt, src, _ = c5
    assert np.array_equal(c5_engine.get_neighbors("target"), O.knn(tgt, 20))
    assert np.array_equal(c5_engine.get_neighbors("source"), O.knn(src, 20))


def test_c5_voxelmap_and_linearize(O, c5, c5_engine):
    tgt, src, T = c5
    c = c5_engine
    cov_t, cov_s = c.get_covariances("target").astype(np.float64), c.get_covariances("source").astype(np.float64)
    coords, num, means, covs = c.get_voxelmap()
    oc, on, om, ocv = O.voxelmap_vgicp(tgt, cov_t, 0.5)
    order_g, order_o = np.lexsort(coords.T), np.lexsort(oc.T)
    assert len(coords) == len(oc) and np.array_equal(coords[order_g], oc[order_o])
    assert np.array_equal(num[order_g], on[order_o]) and int(num.sum()) == len(tgt)
    np.testing.assert_allclose(means[order_g], om[order_o], rtol=0, atol=4e-6 * np.abs(om).max())
    np.testing.assert_allclose(covs[order_g], ocv[order_o], rtol=0, atol=2e-7)
    refs = []
    for rnd, tol in ((True, 1e-9), (False, 1e-5)):
        g = O.FastVGICP(search=D7, resolution=0.5, round_fp32=rnd)
        g.set_target(tgt); g.set_source(src); g.set_target_covs(cov_t); g.set_source_covs(cov_s); g.prepare()
        refs.append((g, tol))
    _check_sums(c, refs, _poses(T, 17), 19)


def test_c5_align(O, c5, c5_engine):
    tgt, src, T = c5
    c = c5_engine
    r = c.align()
    g = O.FastVGICP(search=D7, resolution=0.5)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4 and util.rel_err(r["H"], ro["H"]) < 1e-4
    f, fo = c.fitness_score(r["T"].astype(np.float32).astype(np.float64)), g.fitness()
    assert abs(f - fo) <= 1e-4 * fo
    te, re_ = util.pose_error(T, r["T"])
    assert te < 0.02 and re_ < np.radians(0.1)


# =====================================================================================================
# C4: simulated 64-ring LiDAR frames, NDT
# =====================================================================================================
@pytest.fixture(scope="module")
def c4(O):
    raw = [util.lidar_frame(i) for i in (3, 4)]
    assert min(len(f) for f in raw) > 100_000
    return raw, [O.approx_voxelgrid(f, 0.25) for f in raw]  # kitti.cpp:80-82


def _ndt():
    from fast_gicp_amd import capi
    return capi.NDTCore(0)


@pytest.mark.parametrize("mode", [1, 0])
def test_c4_ndt_frame_pair(O, c4, mode):
    """kitti.cpp:115-128 odometry step: previous frame = target, current frame = source, both downsampled (D2D, the NDTCuda
    default, and P2D)."""
    _, (tgt, src) = c4
    c = _ndt()
    c.set_distance_mode(mode); c.set_neighbor_search_method(D7); c.set_resolution(1.0)
    c.set_target_cloud(tgt); c.set_source_cloud(src); c.create_voxelmaps()
    maps = {}
    for which, cloud in (("target", tgt),) + ((("source", src),) if mode == 1 else ()):
        coords, num, means, covs = c.get_voxelmap(which)
        oc, on, om, ocv = O.voxelmap_ndt(cloud, 1.0)
        og, oo = np.lexsort(coords.T), np.lexsort(oc.T)
        assert len(coords) == len(oc) and np.array_equal(coords[og], oc[oo]) and np.array_equal(num[og], on[oo])
        np.testing.assert_allclose(means[og], om[oo], rtol=0, atol=4e-6 * np.abs(om).max())
        np.testing.assert_allclose(covs[og], ocv[oo], rtol=0, atol=1e-5 * max(1e-3, np.abs(ocv).max()))
        maps[which] = (coords, num, means.astype(np.float64), covs.astype(np.float64))
    g64 = O.NDT(mode=mode, search=D7)
    g64.set_target(tgt); g64.set_source(src); g64.prepare()
    # "oracle fed the same fp32 voxel data": the engine's own records injected -> only the cost arithmetic is compared
    g32 = O.NDT(mode=mode, search=D7)
    g32.set_target(tgt); g32.set_source(src); g32.prepare()
    for which, rec in maps.items():
        g32.set_voxelmap(which, *rec)
    gt = np.linalg.inv(util.lidar_pose(3)) @ util.lidar_pose(4)
    _check_sums(c, [(g32, 1e-9), (g64, 2e-5)], _poses(gt, 23), 29, d2d=(mode == 1))
    r = c.align()
    go = O.NDT(mode=mode, search=D7)
    go.set_target(tgt); go.set_source(src)
    ro = go.align()
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4 and util.rel_err(r["H"], ro["H"]) < 1e-4
    te, re_ = util.pose_error(gt, r["T"])
    assert te < 0.1 and re_ < np.radians(0.5)
    c.close()


def test_c4_ndt_raw_118k_frame_p2d(O, c4):
    """The un-downsampled 118k-point frame as P2D source against the voxel map of the previous raw frame: the large-cloud
    walk of the NDT instantiation of the cost kernel."""
    (tgt, src), _ = c4
    c = _ndt()
    c.set_distance_mode(0); c.set_neighbor_search_method(D7); c.set_resolution(1.0)
    c.set_target_cloud(tgt); c.set_source_cloud(src); c.create_voxelmaps()
    coords, num, means, covs = c.get_voxelmap("target")
    assert int(num.sum()) == len(tgt)
    g32 = O.NDT(mode=0, search=D7)
    g32.set_target(tgt); g32.set_source(src); g32.prepare()
    oc, on, _, _ = g32.get_voxelmap("target")
    og, oo = np.lexsort(coords.T), np.lexsort(oc.T)
    assert np.array_equal(coords[og], oc[oo]) and np.array_equal(num[og], on[oo])
    g32.set_voxelmap("target", coords, num, means.astype(np.float64), covs.astype(np.float64))
    gt = np.linalg.inv(util.lidar_pose(3)) @ util.lidar_pose(4)
    _check_sums(c, [(g32, 1e-9)], _poses(gt, 31)[:2], 37)
    c.close()
