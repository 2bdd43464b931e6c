"""GPU parity tests proper: the HIP engine, called through the C ABI (fast_gicp_amd.capi ->
libfast_vgicp_hip.so), against the oracle on the same inputs.

Tolerances (written here on purpose):
  * integer/index work (voxel sets, point counts, k-NN index sets, correspondence counts): exact;
  * voxel means/covs, covariances: fp32 storage rounding (rel 3e-6 of the entry scale);
  * H, b, error at fixed poses (fp64 math, fp32-stored inputs): rel 1e-5 vs the all-fp64 oracle,
    rel 1e-9 vs the oracle fed the same fp32-rounded inputs;
  * final transform / fitness on the bundled pair: 1e-4 relative (BASELINE.json north_star).
"""
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


@pytest.fixture(scope="module")
def oracle_covs(O, pair):
    tgt, src = pair
    return O.covariances_knn(tgt, 20, O.PLANE), O.covariances_knn(src, 20, O.PLANE)


def _core():
    from fast_gicp_amd import capi
    return capi.VGICPCore(0)


def test_library_loaded_is_in_tree():
    from fast_gicp_amd import capi
    import os
    assert os.path.exists(capi.lib_path())
    assert capi.device_count() >= 1


def test_voxelmap_matches_oracle(O, pair, oracle_covs):
    tgt, _ = pair
    cov_t, _ = oracle_covs
    for res in (1.0, 0.5):
        c = _core()
        c.set_resolution(res)
        c.set_target_cloud(tgt)
        c.set_target_covariances(cov_t)
        c.create_target_voxelmap()
        coords, num, means, covs = c.get_voxelmap()
        oc, on, om, ocv = O.voxelmap_vgicp(tgt, cov_t, res)
        got = util.voxel_dict(coords, num, means, covs)
        ref = util.voxel_dict(oc, on, om, ocv)
        assert set(got) == set(ref), "voxel coordinate sets differ"
        assert len(got) == len(coords), "duplicate voxels in the table"
        assert int(num.sum()) == len(tgt), "points dropped"
        for k in ref:
            assert got[k][0] == ref[k][0]
            np.testing.assert_allclose(got[k][1], ref[k][1], rtol=0, atol=4e-6 * max(1.0, np.abs(ref[k][1]).max()))
            np.testing.assert_allclose(got[k][2], ref[k][2], rtol=0, atol=2e-7)
        c.close()


@pytest.mark.parametrize("search,name", [(2, "DIRECT1"), (1, "DIRECT7"), (0, "DIRECT27")])
def test_linearize_matches_oracle(O, pair, oracle_covs, search, name):
    tgt, src = pair
    cov_t, cov_s = oracle_covs
    c = _core()
    c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.set_target_covariances(cov_t); c.set_source_covariances(cov_s)
    c.create_target_voxelmap()
    # oracle twice: all-fp64, and fed the fp32-rounded covariances / voxel data like the engine stores them
    refs = []
    for rnd in (False, True):
        g = O.FastVGICP(search=search, round_fp32=rnd)
        g.set_target(tgt); g.set_source(src)
        g.set_target_covs(cov_t.astype(np.float32).astype(np.float64) if rnd else cov_t)
        g.set_source_covs(cov_s.astype(np.float32).astype(np.float64) if rnd else cov_s)
        g.prepare()
        refs.append(g)
    poses = [np.eye(4), util.relative_pose(), util.random_pose(np.random.default_rng(7))]
    for T in poses:
        e, H, b = c.linearize(T)
        n_corr = c.get_num_correspondences()
        for g, tol in zip(refs, (1e-5, 1e-9)):
            eo, Ho, bo = g.linearize(T)
            assert n_corr == g.num_correspondences()
            assert abs(e - eo) <= tol * abs(eo)
            assert util.rel_err(H, Ho) <= tol
            assert util.rel_err(b, bo) <= tol
        # trial-step error: correspondences and M stay those of the linearisation pose
        T2 = util.random_pose(np.random.default_rng(11), 0.2, 0.05) @ T
        e2 = c.compute_error(T2, derivatives=False)
        e2o = refs[1].compute_error(T2)
        assert abs(e2 - e2o) <= 1e-9 * abs(e2o)
        e3, H3, b3 = c.compute_error(T2, derivatives=True)
        assert abs(e3 - e2) <= 1e-12 * abs(e2)
    c.close()


def test_knn_bruteforce_sets_equal_oracle(O, pair):
    tgt, src = pair
    c = _core()
    c.set_source_cloud(src)
    c.find_source_neighbors(20)
    got = c.get_neighbors("source")
    ref = O.knn(src, 20)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), "k-NN indices differ (order is ascending (distance, index) on both sides)"
    c.close()


@pytest.mark.parametrize("reg", [3, 1, 4, 2, 0])
def test_covariances_knn_match_oracle(O, pair, reg):
    _, src = pair
    src = src[:6000]
    c = _core()
    c.set_source_cloud(src)
    c.find_source_neighbors(20)
    c.calculate_source_covariances(reg)
    got = c.get_covariances("source").astype(np.float64)
    ref = O.covariances_knn(src, 20, reg)
    if reg in (0, 4):  # NONE / FROBENIUS: no eigenvectors involved -> plain fp32 storage rounding on every point
        scale = np.abs(ref).max(axis=(1, 2))
        assert (np.abs(got - ref).max(axis=(1, 2)) <= 2e-7 * scale + 1e-12).all()
    else:
        # eigen-based: every point against its conditioning-aware bound (fp64 centred sums on both sides: input error 1e-13);
        # points whose eigenvectors are undefined at that precision (exact ties) may only be a handful
        raw = O.covariances_knn(src, 20, O.NONE)
        err, bound, degenerate = util.cov_error_bound(got, ref, raw, input_rel=1e-13, gaps="01" if reg == 3 else "min")
        assert degenerate.sum() <= 5
        assert np.all(err[~degenerate] <= bound[~degenerate]), float((err / bound)[~degenerate].max())
    c.close()


def test_covariances_rbf_match_oracle(O, pair):
    _, src = pair
    src = src[:5000]
    c = _core()
    c.set_kernel_params(0.5, 2.5)
    c.set_source_cloud(src)
    c.calculate_source_covariances_rbf(0)  # raw weighted covariance
    got = c.get_covariances("source").astype(np.float64)
    ref = O.covariances_rbf(src, 0.5, 2.5, 0)
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    err = np.abs(got - ref) / np.maximum(scale, 1e-12)
    assert err.max() < 5e-5, err.max()
    c.calculate_source_covariances_rbf(3)
    got = c.get_covariances("source").astype(np.float64)
    ref = O.covariances_rbf(src, 0.5, 2.5, 3)
    # PLANE depends on the eigenvector of the smallest eigenvalue only: the 5e-5 input error of the fp32 sums is amplified by
    # lambda_max / (lambda_1 - lambda_0); every point is held to that bound, near-degenerate ones are counted
    rawo = O.covariances_rbf(src, 0.5, 2.5, 0)
    err, bound, degenerate = util.cov_error_bound(got, ref, rawo, input_rel=5e-5)
    assert degenerate.sum() <= 5
    assert np.all(err[~degenerate] <= bound[~degenerate]), float((err / bound)[~degenerate].max())
    c.close()


def _engine_register(c, tgt, src, k=20, reg=3):
    c.set_target_cloud(tgt); c.find_target_neighbors(k); c.calculate_target_covariances(reg); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(k); c.calculate_source_covariances(reg)
    return c.align()


@pytest.mark.parametrize("search", [2, 0])
def test_align_bundled_pair_matches_oracle(O, pair, search):
    """north_star: final transform and fitness within 1e-4 relative of the CPU FastVGICP restatement."""
    tgt, src = pair
    c = _core()
    c.set_neighbor_search_method(search)
    r = _engine_register(c, tgt, src)
    g = O.FastVGICP(search=search)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4
    assert util.rel_err(r["H"], ro["H"]) < 1e-4
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    fo = g.fitness()
    assert abs(f - fo) <= 1e-4 * fo
    # reference's own test tolerance against data/relative.txt (gicp_test.cpp:148-149)
    te, re_ = util.pose_error(util.relative_pose(), r["T"])
    assert te < 0.05 and re_ < np.radians(1.0)
    c.close()


def test_fitness_matches_oracle(O, pair):
    tgt, src = pair
    c = _core()
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    for T in (np.eye(4), util.relative_pose()):
        f = c.fitness_score(T)
        fo = O.fitness(src, tgt, T)
        assert abs(f - fo) <= 1e-9 * fo
    c.close()


def test_voxel_table_hint_overflow_falls_back(O, pair, oracle_covs):
    """A too-small capacity hint must cost a rebuild, never points: results identical to the safe-sized table."""
    tgt, src = pair
    cov_t, cov_s = oracle_covs
    ref = None
    for hint in (-1, 1, 100000):
        c = _core()
        c.set_resolution(0.5)  # 2,587 voxels > the 1,024-bucket table a hint of 1 gives
        c.set_target_cloud(tgt); c.set_source_cloud(src)
        c.set_target_covariances(cov_t); c.set_source_covariances(cov_s)
        c.debug_set_voxel_hint(hint)
        c.create_target_voxelmap()
        cap0 = c.debug_table_capacity()
        r = c.align()
        e, H, b = c.linearize(np.eye(4))
        coords, num, _, _ = c.get_voxelmap()
        assert int(num.sum()) == len(tgt) and len(coords) == 2587
        if hint == 1:
            assert cap0 == 1024 and c.debug_table_capacity() == 65536  # overflow detected -> rebuilt at 2 x N_t
        if ref is None:
            ref = (r["T"], e, H)
        else:
            assert util.rel_err(r["T"], ref[0]) < 1e-12 and abs(e - ref[1]) <= 1e-12 * abs(ref[1]) and util.rel_err(H, ref[2]) < 1e-12
        c.close()


def test_sharded_path_single_rank_rccl(O, pair):
    """The multi-GPU code path on the 1-GPU box: a 1-rank RCCL communicator is attached, so every evaluation goes
    cost kernel -> ncclAllReduce(32 doubles, on the handle's stream) -> lm_update kernel.  Must equal the fused path."""
    from fast_gicp_amd import capi, distributed as D
    tgt, src = pair
    a = _core()
    ra = _engine_register(a, tgt, src)
    b = _core()
    sh = D.ShardedVGICP(b, rank=0, world_size=1)
    sh.init_device_collective(capi.comm_unique_id())
    sh.set_target(tgt); sh.set_source(src)
    rb = sh.align()
    assert rb["converged"] and rb["num_linearize"] == ra["num_linearize"]
    assert util.rel_err(rb["T"], ra["T"]) < 1e-12 and util.rel_err(rb["H"], ra["H"]) < 1e-12
    # host-driven variant (all-reduce outside the engine) on the same data
    c = _core()
    sh2 = D.ShardedVGICP(c, rank=0, world_size=1, device_collective=False)
    sh2.set_target(tgt); sh2.set_source(src)
    rc = sh2.align()
    assert util.rel_err(rc["T"], ra["T"]) < 1e-9
    a.close(); b.close(); c.close()


def test_persistent_and_multi_launch_paths_agree(O, pair, monkeypatch):
    """The LM loop runs as ONE persistent launch by default; its barrier watchdog must turn a stuck barrier into a clean
    fallback to one launch per LM transition. Forcing the watchdog (0 ticks) and disabling the persistent kernel must
    both give the bit-identical result of the persistent run (same sums, same fixed summation order)."""
    tgt, src = pair
    c = _core()
    c.set_neighbor_search_method(2)
    r0 = _engine_register(c, tgt, src)
    assert r0["num_launches"] == 1 and c.debug_persist_aborts() == 0
    monkeypatch.setenv("FVH_PERSIST_WATCHDOG_TICKS", "0")
    r1 = c.align()
    assert c.debug_persist_aborts() == 1 and r1["num_launches"] > 1
    monkeypatch.delenv("FVH_PERSIST_WATCHDOG_TICKS")
    rb = c.align()  # back-off: the align right after an abort stays on the multi-launch route (another process may own the GPU)...
    assert c.debug_persist_aborts() == 1 and rb["num_launches"] > 1
    r2 = c.align()  # ... the next one tries the persistent kernel again
    assert c.debug_persist_aborts() == 1 and r2["num_launches"] == 1
    for r in (r1, rb, r2):
        assert r["converged"] and r["num_linearize"] == r0["num_linearize"] and r["num_error_evals"] == r0["num_error_evals"]
        assert np.array_equal(r["T"], r0["T"]) and np.array_equal(r["H"], r0["H"])
    c.close()


@pytest.mark.parametrize("lm", [
    dict(max_iterations=1),                                              # the first accepted trial ends the outer loop
    dict(max_iterations=3),                                              # ... the third one does
    dict(rotation_epsilon=1e-7, transformation_epsilon=1e-7, max_iterations=6),  # not converged by the thresholds when the iterations
                                                                         # run out (tighter ones only compare rounding noise: rho = 0/0)
    dict(rotation_epsilon=1.0, transformation_epsilon=10.0),             # converged by the first proposal
    dict(lm_max_iterations=1),                                           # a rejected trial ends the align ("lm not converged")
    dict(lm_init_lambda_factor=1e3),                                     # heavy damping: many small steps
])
def test_every_way_the_lm_loop_ends_matches_the_oracle(O, pair, monkeypatch, lm):
    """lsq_registration_impl.hpp:53-168. The device loop evaluates a trial whose speculative linearisation can never be used (the
    proposed step is already below the thresholds, or accepting it exhausts max_iterations) on the stored correspondences only
    (PH_TRIAL_FINAL): whatever ends the loop, the bookkeeping (linearisations, error evaluations, iterations, converged) must be
    the oracle's, the transform within 1e-4, and the one-launch and one-launch-per-transition routes bit-identical."""
    tgt, src = pair
    c = _core()
    c.set_neighbor_search_method(1)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    r = c.align(**lm)
    g = O.FastVGICP(search=1)
    okw = dict(lm)
    if "lm_init_lambda_factor" in okw: okw["init_lambda_factor"] = okw.pop("lm_init_lambda_factor")
    g.set_lm(**okw)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] == ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"] and r["iterations"] == ro["iterations"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4
    assert r["num_launches"] == 1
    monkeypatch.setenv("FVH_PERSISTENT", "0")  # (read once per process -- if it already was, the watchdog hook below still forces the other route)
    monkeypatch.setenv("FVH_PERSIST_WATCHDOG_TICKS", "0")
    r1 = c.align(**lm)
    monkeypatch.delenv("FVH_PERSIST_WATCHDOG_TICKS")
    monkeypatch.delenv("FVH_PERSISTENT")
    assert r1["num_launches"] > 1 or r["num_error_evals"] == 0
    assert np.array_equal(r1["T"], r["T"]) and np.array_equal(r1["H"], r["H"]) and r1["num_error_evals"] == r["num_error_evals"]
    c.align()  # (leave the handle's back-off state as the next test expects it)
    c.close()


@pytest.mark.parametrize("n", [17334, 4099, 64, 20])
def test_cooperative_sort_and_its_fallback_give_the_same_knn(O, pair, monkeypatch, n):
    """Small clouds are Morton-sorted by a cooperative kernel (32 workgroups meeting at grid barriers) with a
    single-workgroup kernel behind it that takes over when the barrier watchdog fires. Both routes must feed the exact
    k-NN the same way: identical neighbour lists, equal to the oracle's."""
    _, src = pair
    src = np.ascontiguousarray(src[:n])
    k = min(20, n)
    ref = O.knn(src, k)
    c = _core()  # the only engine alive in this process at this point -> cooperative path
    c.set_source_cloud(src); c.find_source_neighbors(k)
    got = c.get_neighbors("source")
    monkeypatch.setenv("FVH_SORT_COOP_WATCHDOG_TICKS", "0")
    c.set_source_cloud(src); c.find_source_neighbors(k)
    got_fb = c.get_neighbors("source")
    monkeypatch.delenv("FVH_SORT_COOP_WATCHDOG_TICKS")
    assert np.array_equal(got, ref) and np.array_equal(got_fb, ref)
    c.close()


def test_multiplicative_voxel_accumulation_matches_oracle(O, pair, oracle_covs):
    """VoxelAccumulationMode::MULTIPLICATIVE (fast_vgicp_voxel.hpp:79-103, CPU-only in the reference): voxel sets
    and counts exact, means / covariances fp32 rounding, err / H / b 1e-9 against the oracle fed the same fp32-rounded data,
    final transform 1e-4 with equal iteration counts; ADDITIVE_WEIGHTED == ADDITIVE as in the reference."""
    from fast_gicp_amd import capi
    tgt, src = pair
    cov_t, cov_s = oracle_covs
    c = _core()
    c.set_neighbor_search_method(1)
    c.set_voxel_accumulation_mode(capi.VOXEL_MULTIPLICATIVE)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.set_target_covariances(cov_t); c.set_source_covariances(cov_s)
    c.create_target_voxelmap()
    coords, num, means, covs = c.get_voxelmap()
    oc, on, om, ocv = O.voxelmap_vgicp(tgt, cov_t.astype(np.float32).astype(np.float64), 1.0, O.MULTIPLICATIVE)
    og, oo = np.lexsort(coords.T), np.lexsort(oc.T)
    assert len(coords) == len(oc) and np.array_equal(coords[og], oc[oo]) and np.array_equal(num[og], on[oo])
    np.testing.assert_allclose(means[og], om[oo], rtol=0, atol=4e-6 * np.abs(om).max())
    np.testing.assert_allclose(covs[og], ocv[oo], rtol=2e-6, atol=1e-9)
    g = O.FastVGICP(search=O.DIRECT7, round_fp32=True)
    g.set_voxel_accumulation_mode(O.MULTIPLICATIVE)
    g.set_target(tgt); g.set_source(src)
    g.set_target_covs(cov_t.astype(np.float32).astype(np.float64)); g.set_source_covs(cov_s.astype(np.float32).astype(np.float64))
    g.prepare()
    for T in (np.eye(4), util.relative_pose()):
        e, H, b = c.linearize(T)
        eo, Ho, bo = g.linearize(T)
        assert c.get_num_correspondences() == g.num_correspondences()
        assert util.sums_close(e, H, b, eo, Ho, bo, 1e-9), (e, eo, util.rel_err(H, Ho))
    r = c.align()
    ro = g.align()
    assert r["converged"] and ro["converged"] and r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4
    te, re_ = util.pose_error(util.relative_pose(), r["T"])
    assert te < 0.05 and re_ < np.radians(1.0)
    # the mode is part of the map: switching back rebuilds an additive map equal to a fresh handle's
    c.set_voxel_accumulation_mode(capi.VOXEL_ADDITIVE_WEIGHTED)
    c.create_target_voxelmap()
    _, _, m1, v1 = c.get_voxelmap()
    d = _core()
    d.set_target_cloud(tgt); d.set_target_covariances(cov_t); d.create_target_voxelmap()
    _, _, m0, v0 = d.get_voxelmap()
    assert np.array_equal(np.sort(m1, axis=0), np.sort(m0, axis=0)) and np.array_equal(np.sort(v1.reshape(len(v1), -1), axis=0), np.sort(v0.reshape(len(v0), -1), axis=0))
    c.close(); d.close()


@pytest.mark.parametrize("search", ["DIRECT7", "DIRECT27"])
def test_gauss_newton_on_the_device_matches_the_oracle_on_both_routes(O, pair, search):
    """fvh_lm_params::optimizer = 1: LsqRegistration::step_gn (lsq_registration_impl.hpp:108-121) inside the kernel -- every transition a
    linearisation, H d = -b undamped, x0 = exp(d) x0, final_hessian_ = H, converged_ = is_converged(delta). Against the oracle's step_gn
    (pose 1e-4, equal iteration count, no error evaluations), and the persistent launch against one launch per transition: same bits."""
    import os
    import subprocess
    import sys
    from fast_gicp_amd import capi
    tgt, src = pair
    cs, os_ = {"DIRECT7": (capi.DIRECT7, O.DIRECT7), "DIRECT27": (capi.DIRECT27, O.DIRECT27)}[search]
    g = O.FastVGICP(search=os_)
    g.set_optimizer("GN")
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(cs)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
    r = c.align(optimizer=1)
    assert r["converged"] and ro["converged"] and r["num_launches"] == 1
    assert r["num_error_evals"] == 0 == ro["num_error_evals"] and r["num_linearize"] == ro["num_linearize"] and r["iterations"] == ro["iterations"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4 and util.rel_err(r["H"], ro["H"]) < 1e-4
    rl = c.align()  # Levenberg-Marquardt on the same handle afterwards: the optimiser is a per-align parameter
    assert rl["converged"] and rl["num_error_evals"] > 0
    # one launch per transition (FVH_PERSISTENT=0, its own process: the knob is read once): bit-identical to the persistent launch
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from tests import util
from fast_gicp_amd import capi
tgt, src = util.bundled_pair()
c = capi.VGICPCore(0)
c.set_neighbor_search_method(%d)
c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
r = c.align(optimizer=1)
assert r["num_launches"] > 1
np.save(sys.argv[1], np.concatenate([r["T"].ravel(), r["H"].ravel()]))
""" % (util.ROOT, cs)
    import tempfile
    out = os.path.join(tempfile.mkdtemp(), "gn.npy")
    subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, FVH_PERSISTENT="0"), cwd=util.ROOT)
    assert np.array_equal(np.load(out), np.concatenate([r["T"].ravel(), r["H"].ravel()]))
    c.close()
