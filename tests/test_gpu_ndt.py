"""GPU parity of the NDT path (NDTCudaCore replacement) against the oracle's fp64 restatement of
ndt_cuda.cu / ndt_compute_derivatives.cu, through the C ABI.

Tolerances: voxel sets / counts exact; voxel means fp32 rounding; MIN_EIG-regularised covariances
1e-5 of the entry scale (fp32 storage); err/H/b at fixed poses rel 2e-5 vs the all-fp64 oracle and rel 1e-9 vs the oracle fed
the engine's fp32-stored voxel records;
final pose within 1e-4 relative, and the reference's own gicp_test tolerance vs data/relative.txt."""
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


def _ndt():
    from fast_gicp_amd import capi
    return capi.NDTCore(0)


def test_ndt_voxelmaps_match_oracle(O, pair):
    tgt, src = pair
    c = _ndt()
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.create_voxelmaps()
    for which, cloud in (("target", tgt), ("source", src)):
        coords, num, means, covs = c.get_voxelmap(which)
        oc, on, om, ocv = O.voxelmap_ndt(cloud, 1.0)
        got, ref = util.voxel_dict(coords, num, means, covs), util.voxel_dict(oc, on, om, ocv)
        assert set(got) == set(ref)
        assert int(num.sum()) == len(cloud)
        for k in ref:
            assert got[k][0] == ref[k][0]
            np.testing.assert_allclose(got[k][1], ref[k][1], rtol=0, atol=4e-6 * max(1.0, np.abs(ref[k][1]).max()))
            np.testing.assert_allclose(got[k][2], ref[k][2], rtol=0, atol=1e-5 * max(1e-3, np.abs(ref[k][2]).max()))
    c.close()


@pytest.mark.parametrize("mode,search", [(1, 1), (0, 1), (1, 2), (1, 0)])
def test_ndt_linearize_matches_oracle(O, pair, mode, search):
    tgt, src = pair
    c = _ndt()
    c.set_distance_mode(mode); c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.create_voxelmaps()
    g = O.NDT(mode=mode, search=search)
    g.set_target(tgt); g.set_source(src); g.prepare()
    # second leg: the oracle fed the engine's own fp32-stored voxel records -> what is left is the cost arithmetic
    # (ndt_compute_derivatives.cu:104-175), held to the same 1e-9 as the VGICP sums
    g32 = O.NDT(mode=mode, search=search)
    g32.set_target(tgt); g32.set_source(src); g32.prepare()
    for which in ("target",) + (("source",) if mode == 1 else ()):
        coords, num, means, covs = c.get_voxelmap(which)
        g32.set_voxelmap(which, coords, num, means.astype(np.float64), covs.astype(np.float64))
    for T in (np.eye(4), util.relative_pose(), util.random_pose(np.random.default_rng(5))):
        e, H, b = c.linearize(T)
        T2 = util.random_pose(np.random.default_rng(9), 0.2, 0.05) @ T
        e2 = c.compute_error(T2, derivatives=False)
        for ref, tol in ((g, 2e-5), (g32, 1e-9)):
            eo, Ho, bo = ref.linearize(T)
            assert c.get_num_correspondences() == ref.num_correspondences()
            assert abs(e - eo) <= tol * abs(eo)
            assert util.rel_err(H, Ho) <= tol and util.rel_err(b, bo) <= tol
            assert abs(e2 - ref.compute_error(T2)) <= tol * abs(ref.compute_error(T2))
    c.close()


@pytest.fixture(scope="module")
def gicp_test_pair():
    """gicp_test.cpp:55-65 input: exact VoxelGrid leaf 0.2, no origin filter (7,908 / 8,061 points)."""
    return util.bundled_pair(origin_filter=False, leaf=0.2, exact_voxelgrid=True)


@pytest.mark.parametrize("mode", [1, 0])
def test_ndt_align_matches_oracle(O, gicp_test_pair, mode):
    tgt, src = gicp_test_pair
    c = _ndt()
    c.set_distance_mode(mode)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    r = c.align()
    g = O.NDT(mode=mode)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    assert r["converged"] and ro["converged"]
    assert util.rel_err(r["T"], ro["T"]) < 1e-4
    te, re_ = util.pose_error(util.relative_pose(), r["T"])
    # gicp_test.cpp:148-149 tolerance applies to what it tests: NDTCuda defaults = D2D; P2D only gets a sanity band
    assert (te < 0.05 if mode == 1 else te < 0.1) and re_ < np.radians(1.0)
    f, fo = c.fitness_score(r["T"].astype(np.float32).astype(np.float64)), g.fitness()
    assert abs(f - fo) <= 1e-4 * fo
    c.close()


def test_ndt_swap_reuses_voxelmaps(O, gicp_test_pair):
    """NDTCudaCore::swap_source_and_target swaps the maps (ndt_cuda.cu:90-93); registering the reverse
    direction after a swap must equal a fresh reverse registration."""
    tgt, src = gicp_test_pair
    c = _ndt()
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    r1 = c.align()
    c.swap_source_and_target()
    r2 = c.align()
    d = _ndt()
    d.set_target_cloud(src); d.set_source_cloud(tgt)
    r3 = d.align()
    assert util.rel_err(r2["T"], r3["T"]) < 1e-9
    te, re_ = util.pose_error(util.relative_pose(), np.linalg.inv(r2["T"]))
    assert te < 0.05 and re_ < np.radians(1.0)
    c.close(); d.close()


@pytest.mark.parametrize("mode", [0, 1])  # P2D, D2D
@pytest.mark.parametrize("nranks", [2, 3])
def test_ndt_source_tiles_partition_the_cost(mode, nranks):
    """fvh_ndt_set_source_tile (multi-GPU NDT by spatial tile): `nranks` handles on the same full clouds, each evaluating
    its tile -- P2D: a chunk of the source points' Morton order; D2D: a chunk of the source voxels ranked by key (every handle builds its
    own source map: the canonical order makes them cut the same list whatever order their builds left it in). The tiles' partial sums
    must add up to the unsharded evaluation (err, H, b to 1e-11: the same terms in another order), at two poses; and the host-route
    registration over the tiles (ShardedLsq: one all-reduce per evaluation, here a plain sum in this process) equals the unsharded align."""
    from fast_gicp_amd import capi, distributed as D, workloads
    vg = capi.VoxelGrid(0)
    tgt, src = vg.filter(workloads.lidar_frame(3), 0.25), vg.filter(workloads.lidar_frame(4), 0.25)

    def make():
        c = capi.NDTCore(0)
        c.set_distance_mode(mode); c.set_neighbor_search_method(capi.DIRECT7); c.set_resolution(1.0)
        c.set_target_cloud(tgt); c.set_source_cloud(src); c.create_voxelmaps()
        return c

    ref = make()
    tiles = [make() for _ in range(nranks)]
    for r, c in enumerate(tiles):
        c.set_source_tile(r, nranks)
    T1 = util.random_pose(np.random.default_rng(7), max_angle_deg=1.0, max_trans=0.3)
    for T in (np.eye(4), T1):
        e0, H0, b0 = ref.linearize(T)
        parts = [c.linearize(T) for c in tiles]
        e, H, b = sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts)
        assert all(p[0] > 0 for p in parts)  # every tile holds part of the scene
        assert abs(e - e0) <= 1e-11 * abs(e0) and util.rel_err(H, H0) < 1e-11 and util.rel_err(b, b0) < 1e-11, (mode, nranks)
        ee = sum(c.compute_error(T1, derivatives=False) for c in tiles)  # error only, at another pose, on the stored correspondences
        assert abs(ee - ref.compute_error(T1, derivatives=False)) <= 1e-11 * abs(ee)
    # a tiled handle without a communicator must not run the LM loop on partial sums; its correspondence getters are not available
    with pytest.raises(capi.FvhError):
        tiles[0].align()
    with pytest.raises(capi.FvhError):
        tiles[0].get_num_correspondences()
    # the host route: LM on the summed evaluations == the unsharded device LM (same algorithm, sums in another order)
    r0 = ref.align()
    lsq = D.ShardedLsq(lambda T: tuple(sum(x) for x in zip(*[c.linearize(T) for c in tiles])), lambda T: sum(c.compute_error(T, derivatives=False) for c in tiles), lambda v: v)
    r = lsq.align()
    assert r["converged"] and r0["converged"] and util.rel_err(r["T"], r0["T"]) < 1e-9
    # switching the tile off gives the whole cloud back
    tiles[0].set_source_tile(0, 1)
    e1, _, _ = tiles[0].linearize(np.eye(4))
    assert abs(e1 - ref.linearize(np.eye(4))[0]) <= 1e-11 * abs(e1)
    for c in tiles + [ref]:
        c.close()
    vg.close()


def test_tiled_d2d_survives_a_table_overflow_rebuild():
    """a tiled D2D handle whose hint-sized SOURCE table overflows rebuilds both maps at the safe size inside compute_error;
    the rebuild reorders the compact voxel list, so the canonical order every rank cuts its tile from has to be ranked again -- a stale
    permutation of the old list is not a partition of the new one (voxels counted twice / never). Two tiles with a hint of one voxel: their
    partial sums still add up to the unsharded evaluation."""
    from fast_gicp_amd import capi, workloads
    vg = capi.VoxelGrid(0)
    tgt, src = vg.filter(workloads.lidar_frame(3), 0.25), vg.filter(workloads.lidar_frame(4), 0.25)

    def make(hint):
        c = capi.NDTCore(0)
        c.set_distance_mode(capi.NDT_D2D); c.set_neighbor_search_method(capi.DIRECT7); c.set_resolution(1.0)
        c.set_target_cloud(tgt); c.set_source_cloud(src)
        if hint:
            c.debug_set_voxel_hint("source", 1); c.debug_set_voxel_hint("target", 1)  # 1,024 buckets for a few thousand voxels
        c.create_voxelmaps()
        return c

    ref = make(False)
    tiles = [make(True) for _ in range(2)]
    for r, c in enumerate(tiles):
        c.set_source_tile(r, 2)
    T = util.random_pose(np.random.default_rng(11), max_angle_deg=1.0, max_trans=0.3)
    e0, H0, b0 = ref.linearize(T)
    parts = [c.linearize(T) for c in tiles]
    e, H, b = sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts)
    assert all(p[0] > 0 for p in parts)
    assert abs(e - e0) <= 1e-11 * abs(e0) and util.rel_err(H, H0) < 1e-11 and util.rel_err(b, b0) < 1e-11, (abs(e - e0) / abs(e0), util.rel_err(H, H0))
    for c in tiles + [ref]:
        c.close()
    vg.close()


@pytest.mark.parametrize("mode", [0, 1])  # P2D, D2D
def test_ndt_table_overflow_is_answered_by_a_safe_rebuild(mode):
    """Both NDT maps built with a capacity hint of one voxel (1,024 buckets for a few thousand voxels): linearize and align must notice the
    overflow counter, rebuild at the safe size and give the results of a handle that never had a hint. (Round 6: the D2D retry used to read
    the source map's OLD counter set -- zero voxels -- after the rebuild had flipped it.)"""
    from fast_gicp_amd import capi, workloads
    vg = capi.VoxelGrid(0)
    tgt, src = vg.filter(workloads.lidar_frame(3), 0.25), vg.filter(workloads.lidar_frame(4), 0.25)

    def make(hint):
        c = capi.NDTCore(0)
        c.set_distance_mode(mode); c.set_neighbor_search_method(capi.DIRECT7); c.set_resolution(1.0)
        c.set_target_cloud(tgt); c.set_source_cloud(src)
        if hint:
            c.debug_set_voxel_hint("source", 1); c.debug_set_voxel_hint("target", 1)
        return c

    ref, c = make(False), make(True)
    T = util.random_pose(np.random.default_rng(12), max_angle_deg=1.0, max_trans=0.3)
    c.create_voxelmaps(); ref.create_voxelmaps()
    e0, H0, b0 = ref.linearize(T)
    e, H, b = c.linearize(T)
    assert e0 > 0 and abs(e - e0) <= 1e-11 * abs(e0) and util.rel_err(H, H0) < 1e-11 and util.rel_err(b, b0) < 1e-11
    c2 = make(True)
    r0, r = ref.align(), c2.align()  # (align builds the maps itself: with the bad hints)
    assert r["converged"] and r0["converged"] and r["num_linearize"] == r0["num_linearize"] and util.rel_err(r["T"], r0["T"]) < 1e-9
    assert c2.get_num_voxels("target") == ref.get_num_voxels("target")
    for h in (ref, c, c2):
        h.close()
    vg.close()
