"""GPU tests: edge cases of the C ABI, and size-independent properties at BASELINE.json's full sizes (100k / 1M points)
where the oracle cannot finish in seconds."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _core():
    from fast_gicp_amd import capi
    return capi.VGICPCore(0)


# ---------------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------------
def test_error_codes_not_crashes():
    from fast_gicp_amd import capi
    c = _core()
    with pytest.raises(capi.FvhError):
        c.find_source_neighbors(20)            # no cloud yet
    with pytest.raises(capi.FvhError):
        c.create_target_voxelmap()             # no target
    with pytest.raises(capi.FvhError):
        c.align()                              # nothing set
    pts = np.random.default_rng(0).normal(size=(10, 3)).astype(np.float32)
    c.set_source_cloud(pts)
    with pytest.raises(capi.FvhError):
        c.find_source_neighbors(20)            # fewer points than k
    with pytest.raises(capi.FvhError):
        c.find_source_neighbors(0)
    with pytest.raises(capi.FvhError):
        c.find_source_neighbors(65)            # k > 64 unsupported
    with pytest.raises(capi.FvhError):
        c.calculate_source_covariances(7)      # unknown regularisation
    with pytest.raises(capi.FvhError):
        c.set_resolution(0.0)
    with pytest.raises(capi.FvhError):
        c.set_neighbor_search_method(9)
    with pytest.raises(capi.FvhError):
        c.compute_error(np.eye(4))             # no correspondences yet
    c.set_source_cloud(np.zeros((0, 3), np.float32))  # empty cloud is accepted, then refused where it matters
    assert c.num_points("source") == 0
    with pytest.raises(capi.FvhError):
        c.find_source_neighbors(1)
    c.close()


@pytest.mark.parametrize("n", [20, 21, 63, 64, 65, 129, 1000])
def test_small_and_ragged_clouds_knn_and_voxelmap(O, n):
    """Sizes around the 64-point tile and the k boundary, with duplicated points (exact distance ties -> lower index wins)."""
    rng = np.random.default_rng(n)
    pts = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    pts[n // 3] = pts[0]            # exact duplicates
    pts[n // 2] = pts[1]
    c = _core()
    c.set_source_cloud(pts)
    c.find_source_neighbors(20)
    assert np.array_equal(c.get_neighbors("source"), O.knn(pts, 20))
    c.calculate_source_covariances(3)
    c.swap_source_and_target()      # becomes the target: voxel map from covariances
    coords, num, means, covs = c.get_voxelmap()
    oc, on, om, _ = O.voxelmap_vgicp(pts, O.covariances_knn(pts, 20, O.PLANE), 1.0)
    assert set(map(tuple, coords)) == set(map(tuple, oc)) and int(num.sum()) == n
    c.close()


def test_points_on_voxel_boundaries(O):
    """coord = floor(p/res - 0.5): boundaries sit at half-integers; fp64 like the CPU reference."""
    g = np.arange(-3, 4, dtype=np.float64)
    xs = np.concatenate([g + 0.5, g + 0.5 - 1e-7, g + 0.5 + 1e-7, g])
    pts = np.stack(np.meshgrid(xs, xs[:5], xs[:3], indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    cov = np.tile(np.eye(3), (len(pts), 1, 1))
    c = _core()
    c.set_target_cloud(pts); c.set_target_covariances(cov); c.create_target_voxelmap()
    coords, num, _, _ = c.get_voxelmap()
    oc, on, _, _ = O.voxelmap_vgicp(pts, cov, 1.0)
    assert util.voxel_dict(coords, num) == util.voxel_dict(oc, on)
    c.close()


def test_tiny_registration_matches_oracle(O):
    tgt, src, T = util.synthetic_pair(3000, 2500, seed=7, extent=15.0)
    c = _core()
    c.set_neighbor_search_method(1)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    r = c.align()
    g = O.FastVGICP(search=O.DIRECT7)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] == ro["converged"] and util.rel_err(r["T"], ro["T"]) < 1e-4
    te, re_ = util.pose_error(T, r["T"])
    assert te < 0.05 and re_ < np.radians(0.5)
    c.close()


# ---------------------------------------------------------------------------------------------
# full-size properties
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big():
    return util.synthetic_pair(1_000_000, 100_000, seed=44, extent=150.0)


def test_knn_optimality_and_order_invariance_100k(big):
    _, src, _ = big
    c = _core()
    c.set_source_cloud(src); c.find_source_neighbors(20)
    nb = c.get_neighbors("source")
    assert nb.shape == (len(src), 20) and nb.min() >= 0 and nb.max() < len(src)
    assert (nb[:, 0] == np.arange(len(src))).mean() > 0.999          # self first (distance 0)
    rng = np.random.default_rng(1)
    p64 = src.astype(np.float64)
    for q in rng.integers(0, len(src), 40):                           # brute-force check on a sample of queries
        d = ((p64 - p64[q]) ** 2).sum(1)
        kth = np.partition(d, 19)[19]
        assert d[nb[q]].max() <= kth * (1 + 1e-6) + 1e-12
        assert np.all(np.diff(d[nb[q]]) >= -1e-9)                     # ascending
    # permutation invariance: same neighbour SETS (as points) when the input order is shuffled
    perm = rng.permutation(len(src))
    c2 = _core()
    c2.set_source_cloud(src[perm]); c2.find_source_neighbors(20)
    nb2 = c2.get_neighbors("source")
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    sample = rng.integers(0, len(src), 2000)
    d_a = np.sort(((p64[nb[sample]] - p64[sample][:, None]) ** 2).sum(-1), axis=1)
    d_b = np.sort(((p64[perm][nb2[inv[sample]]] - p64[sample][:, None]) ** 2).sum(-1), axis=1)
    assert np.array_equal(d_a, d_b)
    c.close(); c2.close()


def test_voxelmap_conservation_and_order_invariance_1m(big):
    tgt, _, _ = big
    rng = np.random.default_rng(2)
    c = _core()
    c.set_resolution(0.5)
    c.set_target_cloud(tgt); c.calculate_target_covariances_rbf(3); c.create_target_voxelmap()
    coords, num, means, covs = c.get_voxelmap()
    assert int(num.sum()) == len(tgt)                                 # no point dropped
    assert len(np.unique(coords, axis=0)) == len(coords)              # no duplicate voxel
    keys = np.floor(tgt.astype(np.float64) / 0.5 - 0.5).astype(np.int64)
    assert len(np.unique(keys, axis=0)) == len(coords)                # exactly the occupied voxels
    # count-weighted mean of the voxel means = cloud mean (linearity of the accumulation)
    np.testing.assert_allclose((means.astype(np.float64) * num[:, None]).sum(0) / len(tgt), tgt.astype(np.float64).mean(0), atol=2e-5)
    sel = rng.integers(0, len(coords), 200)
    for s in sel:
        m = (keys == coords[s]).all(1)
        assert m.sum() == num[s]
        np.testing.assert_allclose(means[s], tgt[m].astype(np.float64).mean(0), atol=1e-5)
    c.close()


def test_cost_shard_additivity_and_determinism_100k(big):
    """err/H/b of the whole source == sum over spatial tiles (what the multi-GPU all-reduce relies on); repeated
    evaluations are bit-identical (fixed-order reduction)."""
    from fast_gicp_amd import distributed as D
    tgt, src, T = big
    tgt = tgt[:300_000]
    c = _core()
    c.set_resolution(0.5); c.set_neighbor_search_method(1)
    c.set_target_cloud(tgt); c.calculate_target_covariances_rbf(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.calculate_source_covariances_rbf(3)
    covs = c.get_covariances("source").astype(np.float64)
    e, H, b = c.linearize(T)
    e2, H2, b2 = c.linearize(T)
    assert e == e2 and np.array_equal(H, H2) and np.array_equal(b, b2)
    n_all = c.get_num_correspondences()
    tot = np.zeros(43); n_sum = 0
    for tile in D.spatial_tile_partition(src, 4):
        c.set_source_cloud(src[tile]); c.set_source_covariances(covs[tile])
        et, Ht, bt = c.linearize(T)
        tot += np.concatenate([[et], bt, Ht.reshape(-1)])
        n_sum += c.get_num_correspondences()
    assert n_sum == n_all
    assert abs(tot[0] - e) <= 1e-11 * abs(e)
    assert util.rel_err(tot[7:].reshape(6, 6), H) < 1e-11 and util.rel_err(tot[1:7], b) < 1e-11
    c.close()


def test_registration_recovers_ground_truth_1m_map(big):
    """BASELINE configs[4] shape on one GPU: 1M-point map <-> 100k scan, DIRECT7, res 0.5."""
    tgt, src, T = big
    c = _core()
    c.set_resolution(0.5); c.set_neighbor_search_method(1)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    r = c.align()
    assert r["converged"]
    te, re_ = util.pose_error(T, r["T"])
    assert te < 0.02 and re_ < np.radians(0.1)
    # rigid equivariance: moving the scan by a known G must give T * G^-1
    G = util.random_pose(np.random.default_rng(5), 1.0, 0.3)
    src2 = (src.astype(np.float64) @ G[:3, :3].T + G[:3, 3]).astype(np.float32)
    c.set_source_cloud(src2); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    r2 = c.align()
    te2, re2 = util.pose_error(T @ np.linalg.inv(G), r2["T"])
    assert r2["converged"] and te2 < 0.02 and re2 < np.radians(0.1)
    c.close()


def test_full_sweep_modes_agree_with_culled_modes():
    """The un-culled LDS-tiled sweeps (FVH_KNN_MODE/FVH_RBF_MODE/FVH_FIT_MODE=0) of the TEST build of the library
    (-DFVH_TEST_KERNELS, fast_gicp_amd/build.py: the product library no longer carries them) give the culled kernels' results."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from tests import util
from fast_gicp_amd import capi
tgt, src = util.bundled_pair()
src = src[:9000]
c = capi.VGICPCore(0)
c.set_kernel_params(0.5, 2.5)
c.set_target_cloud(tgt); c.set_source_cloud(src)
c.find_source_neighbors(20); nb = c.get_neighbors("source")
c.calculate_source_covariances_rbf(3); cov = c.get_covariances("source")
f = c.fitness_score(util.relative_pose())
np.savez(sys.argv[1], nb=nb, cov=cov, f=f)
''' % util.ROOT
    out = []
    for mode in ("1", "0"):
        from fast_gicp_amd import build as B
        assert os.path.exists(B.TEST_LIB_PATH), "run __graft_entry__.build() (it builds the -DFVH_TEST_KERNELS library too)"
        env = dict(os.environ, FVH_KNN_MODE=mode, FVH_RBF_MODE=mode, FVH_FIT_MODE=mode)
        if mode == "0":
            env["FVH_LIB_PATH"] = B.TEST_LIB_PATH  # mode 1 = the product library
        path = os.path.join(util.ROOT, "gpurun_out", "modes_%s.npz" % mode)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.check_call([sys.executable, "-c", code, path], env=env)
        out.append(np.load(path))
    assert np.array_equal(out[0]["nb"], out[1]["nb"])
    assert np.abs(out[0]["cov"] - out[1]["cov"]).max() < 2e-5   # fp32 partial sums in a different order
    assert abs(float(out[0]["f"]) - float(out[1]["f"])) <= 1e-12 * float(out[1]["f"])


def test_covariance_state_machine_is_consistent(O):
    """The k-NN kernel regularises speculatively with the last-used method; every call order must still give the
    covariances of the method that was actually requested."""
    _, src = util.bundled_pair()
    src = src[:3000]
    c = _core()
    c.set_source_cloud(src)

    def check(method):
        got = c.get_covariances("source").astype(np.float64)
        ref = O.covariances_knn(src, 20, method)
        err = np.abs(got - ref).max(axis=(1, 2)) / np.abs(ref).max(axis=(1, 2))
        assert np.quantile(err, 0.999) < 2e-5, method

    c.find_source_neighbors(20); c.calculate_source_covariances(3); check(3)     # speculation hit (PLANE)
    c.calculate_source_covariances(1); check(1)                                   # different method from the same raw covariances
    c.find_source_neighbors(20); c.calculate_source_covariances(1); check(1)     # speculation now follows MIN_EIG
    c.calculate_source_covariances_rbf(3)
    c.calculate_source_covariances(1); check(1)                                   # RBF overwrote cov: must be recomputed
    c.set_source_covariances(np.tile(np.eye(3), (len(src), 1, 1)))
    c.calculate_source_covariances(4); check(4)
    nb = O.knn(src, 20)
    c.set_source_neighbors(20, nb); c.calculate_source_covariances(3); check(3)  # host-supplied neighbours: gather kernel
    c.close()


@pytest.mark.parametrize("n,search", [(150, 0), (700, 2), (2500, 1), (9000, 0), (40000, 1), (40000, 0)])
def test_persistent_barrier_logic_over_grid_sizes(monkeypatch, n, search):
    """The persistent LM kernel's grid goes from 1 workgroup (fewer groups than arrival counters) to the co-residency cap;
    whatever the size, it must give the bit-identical result of the per-transition launches (forced here through the
    barrier watchdog), for VGICP and both NDT modes."""
    from fast_gicp_amd import capi
    tgt, src, _ = util.synthetic_pair(n, n, seed=7 + n, extent=20.0 if n < 5000 else 60.0)
    c = capi.VGICPCore(0)
    c.set_resolution(1.0); c.set_neighbor_search_method(search)
    k = min(20, n)
    c.set_target_cloud(tgt); c.find_target_neighbors(k); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(k); c.calculate_source_covariances()
    r0 = c.align()
    assert r0["num_launches"] == 1
    monkeypatch.setenv("FVH_PERSIST_WATCHDOG_TICKS", "0")
    r1 = c.align()
    monkeypatch.delenv("FVH_PERSIST_WATCHDOG_TICKS")
    c.align()       # (back-off: one align on the multi-launch route after the abort)
    r2 = c.align()
    assert r2["num_launches"] == 1 and r1["num_launches"] >= 1
    for r in (r1, r2):
        assert r["converged"] == r0["converged"] and r["num_linearize"] == r0["num_linearize"] and r["num_error_evals"] == r0["num_error_evals"]
        assert np.array_equal(r["T"], r0["T"]) and np.array_equal(r["H"], r0["H"])
    c.close()
    for mode in ((1, 0) if n >= 9000 else ()):  # (sparser clouds leave NDT without a voxel of more than 6 points: H = 0)
        d = capi.NDTCore(0)
        d.set_distance_mode(mode); d.set_neighbor_search_method(1)
        d.set_target_cloud(tgt); d.set_source_cloud(src)
        d.align()  # (D2D: the grid of an align is shaped by the source-voxel count the previous align saw; from the second align on it is steady)
        a = d.align()
        monkeypatch.setenv("FVH_PERSIST_WATCHDOG_TICKS", "0")
        b = d.align()
        monkeypatch.delenv("FVH_PERSIST_WATCHDOG_TICKS")
        assert a["num_launches"] == 1
        assert np.isfinite(a["T"]).all() and np.array_equal(a["T"], b["T"]) and a["num_error_evals"] == b["num_error_evals"]
        d.close()


def test_direct_radius_offsets(O):
    """DIRECT_RADIUS (fast_vgicp_cuda.cu:77-91: all integer offsets within radius + 1e-3 of the query voxel). NDT: against the
    oracle, which restates the same offset list; VGICP: radius 1.0 is the DIRECT7 set in another order -> same sums."""
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    for radius in (1.5, 2.0):
        assert len(O.neighbor_offsets(O.DIRECT_RADIUS, radius)) in (19, 33)
        c = capi.NDTCore(0)
        c.set_distance_mode(1); c.set_neighbor_search_method(3, radius)
        c.set_target_cloud(tgt); c.set_source_cloud(src); c.create_voxelmaps()
        g = O.NDT(mode=1, search=O.DIRECT_RADIUS, radius=radius)
        g.set_target(tgt); g.set_source(src); g.prepare()
        for T in (np.eye(4), util.relative_pose()):
            e, H, b = c.linearize(T)
            eo, Ho, bo = g.linearize(T)
            assert c.get_num_correspondences() == g.num_correspondences()
            assert abs(e - eo) <= 2e-5 * abs(eo) and util.rel_err(H, Ho) <= 2e-5 and util.rel_err(b, bo) <= 2e-5
        r, ro = c.align(), g.align()
        assert r["converged"] and ro["converged"] and util.rel_err(r["T"], ro["T"]) < 1e-4
        c.close()
    v = capi.VGICPCore(0)
    v.set_target_cloud(tgt); v.find_target_neighbors(20); v.calculate_target_covariances(); v.create_target_voxelmap()
    v.set_source_cloud(src); v.find_source_neighbors(20); v.calculate_source_covariances()
    v.set_neighbor_search_method(1)  # DIRECT7
    e7, H7, b7 = v.linearize(util.relative_pose())
    r7 = v.align()
    v.set_neighbor_search_method(3, 1.0)
    er, Hr, br = v.linearize(util.relative_pose())
    rr = v.align()
    assert abs(e7 - er) <= 1e-12 * abs(e7) and util.rel_err(Hr, H7) <= 1e-12 and util.rel_err(br, b7) <= 1e-12
    assert util.rel_err(rr["T"], r7["T"]) < 1e-9 and rr["num_error_evals"] == r7["num_error_evals"]
    v.close()


def test_fp32_compute_mode_against_the_cuda_compat_oracle(O):
    """FVH_COMPUTE_FP32 (per-correspondence math in float, sums in double -- the arithmetic class of the reference's CUDA
    kernels) judged against the ORACLE, not against the engine's own fp64 mode: (a) the oracle's cuda-compat leg, i.e.
    compute_derivatives.cu:50-103 restated in float over the same fp32-stored inputs -- 5e-6 (float rounding, different but
    equivalent operation order); (b) the fp64 oracle -- 1e-5 on the sums, 1e-4 on the final transform with equal iteration
    counts; (c) data/relative.txt with the reference's own tolerance."""
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    c = capi.VGICPCore(0)
    c.set_precision(capi.COMPUTE_FP32); c.set_neighbor_search_method(1)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
    g = O.FastVGICP(search=O.DIRECT7, round_fp32=True)
    g.set_target(tgt); g.set_source(src)
    g.set_target_covs(c.get_covariances("target").astype(np.float64)); g.set_source_covs(c.get_covariances("source").astype(np.float64))
    g.prepare()
    for T in (np.eye(4), util.relative_pose()):
        e, H, b = c.linearize(T)
        e64, H64, b64 = g.linearize(T)
        assert c.get_num_correspondences() == g.num_correspondences()
        e32, H32, b32 = g.cuda_compat_sums(T)
        assert util.sums_close(e, H, b, e32, H32, b32, 5e-6), (abs(e - e32) / e32, util.rel_err(H, H32))
        assert util.sums_close(e, H, b, e64, H64, b64, 1e-5), (abs(e - e64) / e64, util.rel_err(H, H64))
        T2 = util.random_pose(np.random.default_rng(3), 0.2, 0.05) @ T
        et = c.compute_error(T2, derivatives=False)
        assert abs(et - g.cuda_compat_sums(T2, derivatives=False)) <= 5e-6 * et
    r32 = c.align()
    ro = g.align()
    assert r32["converged"] and ro["converged"]
    assert r32["num_linearize"] == ro["num_linearize"] and r32["num_error_evals"] == ro["num_error_evals"]
    assert util.rel_err(r32["T"], ro["T"]) < 1e-4
    te, re_ = util.pose_error(util.relative_pose(), r32["T"])
    assert te < 0.05 and re_ < np.radians(1.0)
    c.close()


@pytest.mark.parametrize("k", [21, 32, 33, 40, 64])
def test_covariances_for_large_k_match_oracle(O, k):
    """k_correspondences beyond the reference default: 21..32 take the 8-per-lane register kernel, 33..64 the re-gathering
    kernel (round 1's 16-per-lane instantiation spilled 928 VGPRs and was untested). Exact neighbour lists, covariances to
    fp32 storage rounding (NONE) / the conditioning-aware bound (PLANE)."""
    tgt, _, _ = util.synthetic_pair(6000, 10, seed=11, extent=15.0)
    c = _core()
    c.set_source_cloud(tgt); c.find_source_neighbors(k)
    idx = O.knn(tgt, k)
    assert np.array_equal(c.get_neighbors("source"), idx)
    raw = O.covariances_knn(tgt, k, O.NONE, idx=idx)
    c.calculate_source_covariances(0)
    got = c.get_covariances("source").astype(np.float64)
    assert np.all(np.abs(got - raw).max(axis=(1, 2)) <= 2e-7 * np.abs(raw).max(axis=(1, 2)) + 1e-12)
    c.calculate_source_covariances(3)
    got = c.get_covariances("source").astype(np.float64)
    err, bound, degenerate = util.cov_error_bound(got, O.covariances_knn(tgt, k, O.PLANE, idx=idx), raw, input_rel=1e-13)
    assert degenerate.sum() <= 5 and np.all(err[~degenerate] <= bound[~degenerate])
    c.close()


def test_occupancy_bitmap_changes_nothing_but_the_traffic(tmp_path):
    """Large maps answer the misses of their DIRECT7 / DIRECT27 probes from a cache-resident occupancy bitmap (one bit per voxel
    over the map's bounding box) instead of a key-table sector per probe. Forced on for a small map here (FVH_BITMAP_MIN_POINTS=1),
    and with a budget too small for the box (the grid disables itself and every lookup goes to the table): correspondences, sums
    and the registration must equal the plain lookups' (counts exactly, sums to rounding) in all three cases, for VGICP and NDT."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from tests import util
from fast_gicp_amd import capi
tgt, src, T = util.synthetic_pair(60000, 20000, seed=11, extent=40.0)
src = np.vstack([src, [[500.0, 0.0, 0.0], [np.nan, 0.0, 0.0]]]).astype(np.float32)   # far outside the map's box / not finite
out = {}
for search in (capi.DIRECT7, capi.DIRECT27):
    c = capi.VGICPCore(0)
    c.set_resolution(0.5); c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.set_source_covariances(np.tile(np.eye(3) * 0.01, (len(src), 1, 1)))
    e, H, b = c.linearize(T)
    r = c.align()
    out["v%%d" %% search] = np.concatenate([[e, c.get_num_correspondences()], H.ravel(), b, r["T"].ravel(), [r["num_error_evals"]]])
    c.close()
d = capi.NDTCore(0)
d.set_resolution(1.0); d.set_neighbor_search_method(capi.DIRECT7); d.set_distance_mode(capi.NDT_D2D)
d.set_target_cloud(tgt); d.set_source_cloud(src[:-2])
d.align(); r = d.align()
out["ndt"] = np.concatenate([r["T"].ravel(), [r["num_error_evals"], d.get_num_correspondences()]])
np.savez(sys.argv[1], **out)
""" % util.ROOT
    res = []
    for name, env in (("plain", {"FVH_BITMAP_MIN_POINTS": "100000000"}), ("bitmap", {"FVH_BITMAP_MIN_POINTS": "1"}), ("over_budget", {"FVH_BITMAP_MIN_POINTS": "1", "FVH_BITMAP_MAX_BYTES": "64"})):
        path = str(tmp_path / (name + ".npz"))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, **env))
        res.append(np.load(path))
    # (three processes = three BUILDS of the maps: their fp64 atomics land in another order, and NDT D2D walks its source voxels in
    # arrival order -- sums agree to rounding, not to the bit; every COUNT must be exact)
    for k in res[0].files:
        for other, name in ((res[1], "bitmap"), (res[2], "over budget")):
            a, b = res[0][k], other[k]
            assert util.rel_err(a, b) < 1e-9, (k, name, util.rel_err(a, b))
            counts = [1, -1] if k.startswith("v") else [-2, -1]   # correspondences / error evaluations
            assert all(a[i] == b[i] for i in counts), (k, name)
    assert res[0]["v1"][1] > 10000  # (the comparison is not vacuous: tens of thousands of correspondences)


def test_both_knn_walks_give_the_same_lists(tmp_path):
    """knn_tiled1_kernel<true> (boxes nearest first: clouds up to 65,536 points) and <false> (index order: larger clouds) differ in
    the order candidates are met, never in the result -- the k smallest (distance, index) keys. Each walk is forced onto the size
    the other one normally serves (FVH_KNN_NEAREST_FIRST_MAX_POINTS, read once per process); the lists must be equal element by
    element, for a small cloud with far outliers (the case the nearest-first walk exists for) and for a 100k one."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from fast_gicp_amd import capi
rng = np.random.default_rng(4)
small = rng.uniform(-20, 20, size=(12000, 3)).astype(np.float32); small[:, 2] *= 0.1
small = np.vstack([small, rng.uniform(-400, 400, size=(37, 3)).astype(np.float32)])   # stragglers: their neighbours are far from their tile
big = rng.uniform(-60, 60, size=(100000, 3)).astype(np.float32); big[:, 2] *= 0.05
out = {}
for name, cloud, k in (("small", small, 20), ("small_k64", small[:3000], 64), ("big", big, 20)):
    c = capi.VGICPCore(0)
    c.set_source_cloud(cloud); c.find_source_neighbors(k)
    out[name] = c.get_neighbors("source")
    c.close()
np.savez(sys.argv[1], **out)
""" % util.ROOT
    res = []
    for name, limit in (("index_order", "0"), ("nearest_first", "100000000")):
        path = str(tmp_path / (name + ".npz"))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, FVH_KNN_NEAREST_FIRST_MAX_POINTS=limit))
        res.append(np.load(path))
    for key in res[0].files:
        assert np.array_equal(res[0][key], res[1][key]), key
