"""FastGICP on the device (nearest-target-point correspondences, fast_gicp_impl.hpp:118-240) against the
oracle's restatement of the reference's CPU class, through the C ABI.

Bar: correspondence lists identical (exact 1-NN, fp32 distances, equal distances -> lower index); err/H/b at fixed poses
rel 1e-5 (fp32-stored covariances vs the oracle's fp64 ones; 1e-9 when the oracle is fed the engine's covariances);
final transform and fitness within 1e-4 relative; the reference's own gicp_test tolerance vs data/relative.txt."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def pair():
    return util.bundled_pair()


def _prepared(tgt, src, max_dist=None):
    from fast_gicp_amd import capi
    c = capi.VGICPCore(0)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
    if max_dist is not None:
        c.gicp_set_max_correspondence_distance(max_dist)
    return c


@pytest.mark.parametrize("max_dist", [None, 1.0, 0.3])
def test_correspondences_and_sums_match_oracle(O, pair, max_dist):
    tgt, src = pair
    c = _prepared(tgt, src, max_dist)
    g = O.FastVGICP(k=20)
    g.set_gicp_mode(True, 3.4028234663852886e38 if max_dist is None else max_dist)
    g.set_target(tgt); g.set_source(src)
    # feed the oracle the engine's (fp32-stored) covariances: what is compared is the GICP path, not the covariance estimation
    g.set_target_covs(c.get_covariances("target").astype(np.float64)); g.set_source_covs(c.get_covariances("source").astype(np.float64))
    for T in (np.eye(4), util.relative_pose(), util.random_pose(np.random.default_rng(11))):
        e, H, b = c.gicp_linearize(T)
        eo, Ho, bo = g.linearize(T)
        corr = c.gicp_get_correspondences()
        assert int((corr >= 0).sum()) == g.num_correspondences()
        # exact nearest neighbour (fp32 distances as the reference's float kd-tree query): recompute on the host
        Tf = T.astype(np.float32)
        q = (src[:, 0:1] * Tf[:3, 0] + src[:, 1:2] * Tf[:3, 1]) + (src[:, 2:3] * Tf[:3, 2] + Tf[:3, 3])
        idx, sq = O.knn_query(tgt, q.astype(np.float32), 1)
        thr = np.inf if max_dist is None else float(max_dist) ** 2
        expect = np.where(sq[:, 0].astype(np.float64) < thr, idx[:, 0], -1)
        assert np.array_equal(corr, expect)
        # ... and the oracle's own (source point, target point) list (fast_gicp_impl.hpp:118-156), pair by pair
        assert np.array_equal(np.stack([np.nonzero(corr >= 0)[0], corr[corr >= 0]], axis=1), util.sort_rows(g.correspondences()[:, [0, 4]]))
        assert abs(e - eo) <= 1e-9 * abs(eo)
        assert util.rel_err(H, Ho) <= 1e-9 and util.rel_err(b, bo) <= 1e-9
        T2 = util.random_pose(np.random.default_rng(3), 0.2, 0.05) @ T
        e2 = c.gicp_compute_error(T2, derivatives=False)
        assert abs(e2 - g.compute_error(T2)) <= 1e-9 * abs(g.compute_error(T2))
    c.close()


def test_align_matches_oracle_host_lm(O, pair):
    """The reference's LM loop (LsqRegistration::step_lm) driven from the host over the device cost, vs the oracle's
    FastGICP with its OWN (fp64, CPU-semantics) covariances: north_star tolerance 1e-4 on transform and fitness."""
    from fast_gicp_amd import distributed as D
    tgt, src = pair
    c = _prepared(tgt, src)
    lsq = D.ShardedLsq(lambda T: c.gicp_linearize(T), lambda T: c.gicp_compute_error(T, derivatives=False), lambda v: v)
    r = lsq.align()
    g = O.FastVGICP(k=20)
    g.set_gicp_mode(True)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] and ro["converged"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert abs(f - g.fitness()) <= 1e-4 * g.fitness()
    te, re_ = util.pose_error(util.relative_pose(), r["T"])
    assert te < 0.05 and re_ < np.radians(1.0)  # gicp_test.cpp:148-149
    c.close()


def test_pygicp_fastgicp_class(O, pair):
    """pygicp.FastGICP (main.cpp:183-190) and align_points(method='GICP') on the device engine."""
    import torch  # noqa: F401  (its HIP runtime first, see tests/conftest.py)
    import pygicp
    tgt, src = pair
    g = O.FastVGICP(k=20); g.set_gicp_mode(True, 1.0); g.set_target(tgt); g.set_source(src)
    ro = g.align()
    reg = pygicp.FastGICP()
    reg.set_max_correspondence_distance(1.0)
    reg.set_input_target(tgt); reg.set_input_source(src)
    T = reg.align()
    assert reg.has_converged()
    assert util.rel_err(T, ro["T"]) < 1e-4
    assert abs(reg.get_fitness_score() - g.fitness()) <= 1e-4 * g.fitness()
    T2 = pygicp.align_points(tgt, src, method="GICP", max_correspondence_distance=1.0, k_correspondences=20)
    assert util.rel_err(T2, ro["T"]) < 1e-4
    # swap: registering the reverse direction equals a fresh reverse registration
    reg.swap_source_and_target()
    Tr = reg.align()
    te, re_ = util.pose_error(util.relative_pose(), np.linalg.inv(Tr))
    assert te < 0.05 and re_ < np.radians(1.0)


def test_downsample_device_binding(O):
    import torch  # noqa: F401
    import pygicp
    raw = O.load_pcd(__import__("os").path.join(util.DATA, "251370668.pcd"))
    a = pygicp.downsample(raw, 0.1)
    b = pygicp.downsample_device(raw, 0.1)
    assert a.shape == b.shape == (17249, 3) and np.array_equal(a, b)
    assert np.array_equal(pygicp.downsample_device(raw, 0.2, exact=True).astype(np.float32), O.voxelgrid(raw, 0.2))


@pytest.mark.parametrize("max_dist", [None, 1.0])
def test_device_lm_align_matches_oracle_and_the_host_loop(O, pair, max_dist):
    """fvh_vgicp_gicp_align: the whole FastGICP LM loop on the device (one nearest-point search + one cost launch per LM
    transition, the search reading the next pose from the LM state on the device). Against the oracle's FastGICP: final
    transform / Hessian / fitness 1e-4 with EQUAL linearisation and error-evaluation counts; against the host-driven loop over
    the same device kernels: 1e-9; and the correspondences it leaves behind are those of its last linearisation."""
    from fast_gicp_amd import distributed as D
    tgt, src = pair
    c = _prepared(tgt, src, max_dist)
    r = c.gicp_align()
    g = O.FastVGICP(k=20)
    g.set_gicp_mode(True, 3.4028234663852886e38 if max_dist is None else max_dist)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4 and util.rel_err(r["H"], ro["H"]) < 1e-4
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert abs(f - g.fitness()) <= 1e-4 * g.fitness()
    te, re_ = util.pose_error(util.relative_pose(), r["T"])
    assert te < 0.05 and re_ < np.radians(1.0)  # gicp_test.cpp:148-149
    corr_after = c.gicp_get_correspondences()
    assert int((corr_after >= 0).sum()) > 0.8 * len(src)
    e_after = c.gicp_compute_error(r["T"], derivatives=False)   # legal: nearest-point ids of the last linearisation
    assert np.isfinite(e_after) and e_after > 0
    lsq = D.ShardedLsq(lambda T: c.gicp_linearize(T), lambda T: c.gicp_compute_error(T, derivatives=False), lambda v: v)
    rh = lsq.align()
    assert util.rel_err(r["T"], rh["T"]) < 1e-9
    # a guess near the solution: still the same answer
    r2 = c.gicp_align(util.relative_pose())
    assert r2["converged"] and util.rel_err(r2["T"], r["T"]) < 1e-3
    c.close()


@pytest.mark.parametrize("ns, nt", [(1, 1), (15, 64), (17, 65), (63, 64), (65, 129), (5000, 7001), (20000, 100000), (3000, 300000),
                                    (2000, 1100000)])  # (more than 256 super tiles: the chunk loop of the box walk)
def test_row_per_query_nearest_neighbour_search_is_exact(O, ns, nt):
    """nn1_rows_kernel (round 6: four queries per wave, one per 16-lane row, boxes visited nearest first) against the definition: the nearest
    target point of every transformed source point in the total order (fp32 (dx dx + dy dy) + dz dz without contraction, ORIGINAL index) --
    the oracle's kd-tree query. Ragged sizes (rows, tiles and super tiles that end mid-way; one point; more than 16 super tiles), duplicate
    target points (ties -> lower index), source points at infinity / NaN / far enough to overflow the squared distance (no correspondence,
    excluded from the fitness mean)."""
    from fast_gicp_amd import capi
    rng = np.random.default_rng(ns * 131 + nt)
    tgt = (rng.normal(size=(nt, 3)) * np.array([30.0, 30.0, 3.0])).astype(np.float32)
    if nt >= 129:
        tgt[nt // 2:nt // 2 + 40] = tgt[5:45]          # exact duplicates far apart in index: ties
    src = (rng.normal(size=(ns, 3)) * np.array([30.0, 30.0, 3.0])).astype(np.float32)
    if ns >= 65:
        src[7] = tgt[9]                                 # distance 0 to two target points (9 and its duplicate): the lower index wins
        src[20] = [np.nan, 0.0, 0.0]
        src[21] = [np.inf, 0.0, 0.0]
        src[64] = [1e30, 0.0, 0.0]                      # finite, squared distance overflows to inf
    T = util.random_pose(np.random.default_rng(3), max_angle_deg=3.0, max_trans=1.0)
    if ns >= 65:
        T = np.eye(4)                                   # (the exact-duplicate query must stay exactly on its target point)
    Tf = T.astype(np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        q = ((src[:, 0:1] * Tf[:3, 0] + src[:, 1:2] * Tf[:3, 1]) + (src[:, 2:3] * Tf[:3, 2] + Tf[:3, 3])).astype(np.float32)
    ok = np.isfinite(q).all(1)
    expect = np.full(ns, -1, np.int64)
    best = np.full(ns, np.inf, np.float32)
    if ok.any():
        idx, sq = O.knn_query(tgt, q[ok], 1)
        expect[ok] = np.where(np.isfinite(sq[:, 0]), idx[:, 0], -1)
        best[ok] = sq[:, 0]
    fit_expect = float(np.sum(best[np.isfinite(best)].astype(np.float64)) / max(1, int(np.isfinite(best).sum())))
    c = capi.VGICPCore(0)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.set_target_covariances(np.tile(np.eye(3) * 0.01, (nt, 1, 1))); c.set_source_covariances(np.tile(np.eye(3) * 0.01, (ns, 1, 1)))
    c.gicp_update_correspondences(T)
    corr = c.gicp_get_correspondences()
    assert np.array_equal(corr, expect), (int((corr != expect).sum()), np.nonzero(corr != expect)[0][:10])
    if ns >= 65:
        assert corr[7] == 9
    fit = c.fitness_score(T)
    assert abs(fit - fit_expect) <= 1e-6 * max(fit_expect, 1e-30), (fit, fit_expect)
    c.close()
