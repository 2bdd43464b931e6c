"""GPU tests: the engine's FVH_COMPUTE_FP32 mode END TO END against the oracle's "cuda-compat" leg -- the float restatement of
the reference's device path (oracle/cuda_compat.cpp: FastVGICPCuda / NDTCuda as the .cu files compute them).

north_star asks for parity with FastVGICP *and* FastVGICPCuda. The two reference paths differ from EACH OTHER by 1e-4..3e-4 of
the pose on the bundled pair (tests/test_oracle.py::test_cuda_compat_recorded_values): fp64 centred covariances + fp64 voxel sums
+ fp64 cost on the CPU, float uncentred covariances + float voxel sums + float cost on the device. The engine's fp32 mode keeps
fp64 for everything that is computed ONCE per cloud (covariances, voxel sums, voxel coordinates) and runs the per-correspondence
cost in float with fp64 sums -- it sits between the two reference paths. Tolerances below are stated against that spread:
pose 5e-4 relative, fitness 2e-3 relative, equal iteration counts. At a FIXED pose the cost sums differ by more than the poses do
(err 1e-3, H and b 5e-2): the device path's UNCENTRED float covariances lose |p|^2 / |C| digits -- at 50 m from the sensor the smallest
eigenvector of a 1e-3 m^2 covariance is noise-dominated -- so its Mahalanobis matrices differ from the fp64-centred ones by percents on far
points; the optimum barely moves because near points carry most of the weight."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _engine(tgt, src, search, precision):
    from fast_gicp_amd import capi
    c = capi.VGICPCore(0)
    c.set_precision(precision)
    c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
    return c


@pytest.mark.parametrize("search", ["DIRECT1", "DIRECT7", "DIRECT27"])
def test_vgicp_fp32_mode_end_to_end_vs_cuda_compat(O, search):
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    cs, os_ = {"DIRECT1": (capi.DIRECT1, O.DIRECT1), "DIRECT7": (capi.DIRECT7, O.DIRECT7), "DIRECT27": (capi.DIRECT27, O.DIRECT27)}[search]
    g = O.CudaCompatVGICP(search=os_)
    g.set_target(tgt); g.set_source(src)
    ro = g.align()
    fo = g.fitness()
    c = _engine(tgt, src, cs, capi.COMPUTE_FP32)
    r = c.align()
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"], (r, ro)
    dT = util.rel_err(r["T"], ro["T"])
    print("%s: fp32 engine vs cuda-compat oracle: pose rel %.2e, fitness %.6f vs %.6f" % (search, dT, f, fo))
    assert dT < 5e-4, dT
    assert abs(f - fo) < 2e-3 * fo, (f, fo)
    # the voxel sets are the same whichever arithmetic builds them, and the first correspondence list (identity pose) has the same length
    coords, num, _, _ = g.get_voxelmap()
    ec, en, _, _ = c.get_voxelmap()
    assert util.voxel_dict(ec, en) == util.voxel_dict(coords, num)
    c.update_correspondences(np.eye(4))
    assert c.get_num_correspondences() == g.corr_history()[0]
    # the engine's two precisions against each other: float cost terms move the pose by less than the reference's own CPU / GPU spread
    c64 = _engine(tgt, src, cs, capi.COMPUTE_FP64)
    r64 = c64.align()
    assert util.rel_err(r["T"], r64["T"]) < 2e-4
    c.close(); c64.close()


@pytest.mark.parametrize("search", ["DIRECT1", "DIRECT7", "DIRECT27"])
def test_vgicp_fp32_mode_on_the_cuda_compat_legs_own_covariances(O, search):
    """Which part of the 5e-4 above is covariance noise and which is cost arithmetic: the engine's fp32 mode is fed the cuda-compat leg's OWN
    float covariances (fvh_vgicp_set_*_covariances), so that everything upstream of the voxel map is identical -- what is left is the voxel
    sums (fp64 here, float there: gaussian_voxelmap.cu:164-193), the float voxel coordinate (vector3_hash.cuh:35-38), the float cost terms
    (compute_derivatives.cu:50-135: a different but equivalent order of float operations) and their summation (fp64 here, a float tree
    there). Held to north_star's 1e-4 END TO END with equal iteration counts, identical correspondence lists at every step, and the sums
    at a fixed pose to 1e-4 of their scale. The covariance ESTIMATION (fp64 centred here, float uncentred there) is what the other
    4e-4 of the test above are."""
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    cs, os_ = {"DIRECT1": (capi.DIRECT1, O.DIRECT1), "DIRECT7": (capi.DIRECT7, O.DIRECT7), "DIRECT27": (capi.DIRECT27, O.DIRECT27)}[search]
    g = O.CudaCompatVGICP(search=os_)
    g.set_target(tgt); g.set_source(src); g.prepare()
    T0 = util.relative_pose()
    eo, Ho, bo = g.linearize(T0)  # (estimates the covariances and builds the voxel map on first use)
    c = capi.VGICPCore(0)
    c.set_precision(capi.COMPUTE_FP32)
    c.set_neighbor_search_method(cs)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    c.set_target_covariances(g.get_covs("target")); c.set_source_covariances(g.get_covs("source"))
    c.create_target_voxelmap()
    e, H, b = c.linearize(T0)
    util.assert_same_correspondences(c, g, ordered=True)
    de, dH = abs(e - eo) / abs(eo), util.rel_err(H, Ho)
    db = float(np.max(np.abs(b - bo) / np.sqrt(np.abs(np.diag(Ho)) * abs(eo))))
    print("%s fixed pose, same float covariances: err %.2e H %.2e b %.2e (Cauchy-Schwarz scale)" % (search, de, dH, db))
    assert de < 1e-4 and dH < 1e-4 and db < 1e-4, (de, dH, db)
    ro = g.align()
    fo = g.fitness()
    r = c.align()
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"], (r, ro)
    util.assert_same_correspondences(c, g, ordered=True)  # the lists of the last linearisation of either align
    dT = util.rel_err(r["T"], ro["T"])
    print("%s end to end, same float covariances: pose rel %.2e, fitness %.6f vs %.6f, final H %.2e" % (search, dT, f, fo, util.rel_err(r["H"], ro["H"])))
    assert dT < 1e-4, dT
    assert abs(f - fo) < 1e-4 * fo, (f, fo)
    assert util.rel_err(r["H"], ro["H"]) < 1e-3
    c.close()


def test_vgicp_fp32_cost_sums_at_a_fixed_pose_vs_cuda_compat(O):
    """err / H / b at the ground-truth pose of the bundled pair: the engine's float cost on its own (fp64-built) covariances and voxels
    against the all-float device restatement. What differs is upstream of the cost (uncentred float covariances, float voxel sums):
    err 1e-3, H 5e-2, b 5e-2 of its Cauchy-Schwarz scale (see the module docstring: the float covariances of far points), equal counts."""
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    T = util.relative_pose()
    g = O.CudaCompatVGICP(search=O.DIRECT7)
    g.set_target(tgt); g.set_source(src); g.prepare()
    eo, Ho, bo = g.linearize(T)
    c = _engine(tgt, src, capi.DIRECT7, capi.COMPUTE_FP32)
    e, H, b = c.linearize(T)
    assert c.get_num_correspondences() == g.num_correspondences()
    assert abs(e - eo) < 1e-3 * abs(eo), (e, eo)
    assert util.rel_err(H, Ho) < 5e-2, util.rel_err(H, Ho)
    assert np.all(np.abs(b - bo) <= 5e-2 * np.sqrt(np.abs(np.diag(Ho)) * abs(eo)))
    # the same comparison against the fp64 CPU class bounds what "upstream" means: the engine's float cost on fp64-built inputs is 100x closer to it
    f = O.FastVGICP(search=O.DIRECT7)
    f.set_target(tgt); f.set_source(src); f.prepare()
    ef, Hf, bf = f.linearize(T)
    assert util.sums_close(e, H, b, ef, Hf, bf, 2e-4), (e, ef, util.rel_err(H, Hf))
    c.close()


@pytest.mark.parametrize("mode", ["D2D", "P2D"])
def test_ndt_fp32_mode_end_to_end_vs_cuda_compat(O, mode):
    """NDTCuda (ndt_cuda.cu, ndt_compute_derivatives.cu) on the gicp_test.cpp input (exact VoxelGrid 0.2) and on a pair of simulated
    LiDAR frames: the engine's fp32 mode against the float restatement -- same voxel sets, same iteration counts, pose 1e-3
    (the NDT cost is flatter than VGICP's: float noise in H moves the optimum further), fitness 2e-3."""
    from fast_gicp_amd import capi, workloads
    om, cm = {"D2D": (O.D2D, capi.NDT_D2D), "P2D": (O.P2D, capi.NDT_P2D)}[mode]
    t, s = util.bundled_pair(origin_filter=False, leaf=0.2, exact_voxelgrid=True)
    f0 = O.approx_voxelgrid(workloads.lidar_frame(2), 0.25)
    f1 = O.approx_voxelgrid(workloads.lidar_frame(3), 0.25)
    for name, tgt, src in (("gicp_test pair", t, s), ("lidar frames", f0, f1)):
        g = O.CudaCompatNDT(mode=om, search=O.DIRECT7)
        g.set_target(tgt); g.set_source(src)
        ro = g.align()
        d = capi.NDTCore(0)
        d.set_precision(capi.COMPUTE_FP32)
        d.set_distance_mode(cm); d.set_neighbor_search_method(capi.DIRECT7); d.set_resolution(1.0)
        d.set_target_cloud(tgt); d.set_source_cloud(src)
        r = d.align()
        assert r["converged"] and ro["converged"], name
        assert r["num_linearize"] == ro["num_linearize"], (name, r["num_linearize"], ro["num_linearize"])
        dT = util.rel_err(r["T"], ro["T"])
        fit, fo = d.fitness_score(r["T"].astype(np.float32).astype(np.float64)), g.fitness()
        print("NDT %s, %s: fp32 engine vs cuda-compat oracle: pose rel %.2e, fitness %.6f vs %.6f" % (mode, name, dT, fit, fo))
        assert dT < 1e-3, (name, dT)
        assert abs(fit - fo) < 2e-3 * fo, (name, fit, fo)
        cc, cn, _, _ = g.get_voxelmap("target")
        ec, en, _, _ = d.get_voxelmap("target")
        assert util.voxel_dict(cc, cn) == util.voxel_dict(ec, en), name
        d.close()


def _engine_covs(c, which):
    return c.get_covariances(which).astype(np.float64)


@pytest.mark.parametrize("search", ["DIRECT1", "DIRECT7", "DIRECT27"])
def test_vgicp_cuda_compat_mode_end_to_end_on_its_own_covariances(O, search):
    """north_star's 1e-4 against FastVGICPCuda WITHOUT injecting the oracle's covariances. FVH_COMPUTE_CUDA_COMPAT estimates
    the k-NN covariances as the reference's device path does (covariance_estimation.cu:20-35: uncentred float sums in neighbour-list
    order; covariance_regularization.cu:34-52: Eigen's closed-form float eigen solver, V diag(1e-3, 1, 1) V^-1) -- same association of
    every float operation as oracle/cuda_compat.cpp, no fma contraction -- and runs the float cost on them. Covariances agree with the
    oracle's to float rounding of the trigonometric eigen solver; pose and fitness to 1e-4 with equal iteration counts and identical
    correspondence lists. (The voxel sums stay fp64 here and float there, gaussian_voxelmap.cu:164-193: below 1e-4, as the injected-
    covariance test above already shows.)"""
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    cs, os_ = {"DIRECT1": (capi.DIRECT1, O.DIRECT1), "DIRECT7": (capi.DIRECT7, O.DIRECT7), "DIRECT27": (capi.DIRECT27, O.DIRECT27)}[search]
    g = O.CudaCompatVGICP(search=os_)
    g.set_target(tgt); g.set_source(src); g.prepare()
    c = _engine(tgt, src, cs, capi.COMPUTE_CUDA_COMPAT)
    for which in ("target", "source"):
        a, b = _engine_covs(c, which), g.get_covs(which)
        b6 = np.stack([b[:, 0, 0], b[:, 0, 1], b[:, 0, 2], b[:, 1, 1], b[:, 1, 2], b[:, 2, 2]], 1)
        a6 = np.stack([a[:, 0, 0], a[:, 0, 1], a[:, 0, 2], a[:, 1, 1], a[:, 1, 2], a[:, 2, 2]], 1)
        d = np.abs(a6 - b6).max(1)  # PLANE: eigenvalues (1e-3, 1, 1) -> entries of order 1
        print("%s %s covariances vs cuda-compat oracle: max |d| %.2e, 99.9th percentile %.2e, exactly equal %.1f %%" % (search, which, d.max(), np.percentile(d, 99.9), 100.0 * np.mean(d == 0)))
        # the same float sums bit for bit; what differs is atan2f / cosf / sinf of the closed-form roots (device libm vs glibc: last-ulp),
        # amplified by 1 / (eigenvalue gap) on near-degenerate neighbourhoods
        assert np.percentile(d, 99) < 1e-5 and np.mean(d < 1e-4) > 0.999, (d.max(), np.percentile(d, 99))
    T0 = util.relative_pose()
    eo, Ho, bo = g.linearize(T0)
    e, H, b = c.linearize(T0)
    util.assert_same_correspondences(c, g, ordered=True)
    assert abs(e - eo) < 1e-4 * abs(eo) and util.rel_err(H, Ho) < 1e-4
    ro, r = g.align(), c.align()
    fo = g.fitness()
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"], (r, ro)
    util.assert_same_correspondences(c, g, ordered=True)
    dT = util.rel_err(r["T"], ro["T"])
    print("%s CUDA_COMPAT mode end to end: pose rel %.2e, fitness %.6f vs %.6f" % (search, dT, f, fo))
    assert dT < 1e-4, dT
    assert abs(f - fo) < 1e-4 * fo, (f, fo)
    # and the mode is what separates the two reference paths: the fp64 engine on the same pair differs from it by more than that
    c64 = _engine(tgt, src, cs, capi.COMPUTE_FP64)
    assert 0 < util.rel_err(r["T"], c64.align()["T"]) < 1e-3
    c.close(); c64.close()


# ---- round 6: the two remaining covariance modes of FastVGICPCuda and NDTCuda in the CUDA classes' own arithmetic (kernels_compat.hpp) ----
def _cov6(a):
    a = np.asarray(a, np.float64)
    return np.stack([a[:, 0, 0], a[:, 0, 1], a[:, 0, 2], a[:, 1, 1], a[:, 1, 2], a[:, 2, 2]], 1)


def _rbf_engine(tgt, src, search, precision, resolution=1.0):
    from fast_gicp_amd import capi
    c = capi.VGICPCore(0)
    c.set_precision(precision)
    c.set_resolution(resolution)
    c.set_neighbor_search_method(search)
    c.set_kernel_params(0.5, 3.0)  # FastVGICPCuda's defaults (fast_vgicp_cuda_impl.hpp:31), the oracle leg's too
    c.set_target_cloud(tgt); c.calculate_target_covariances_rbf(capi.REG_PLANE); c.create_target_voxelmap()
    c.set_source_cloud(src); c.calculate_source_covariances_rbf(capi.REG_PLANE)
    return c


def _check_rbf_compat(O, tgt, src, cs, os_, resolution, name):
    from fast_gicp_amd import capi
    g = O.CudaCompatVGICP(search=os_, cov_mode=1, resolution=resolution)
    g.set_target(tgt); g.set_source(src); g.prepare()
    c = _rbf_engine(tgt, src, cs, capi.COMPUTE_CUDA_COMPAT, resolution)
    worst = {}
    for which in ("target", "source"):
        a6, b6 = _cov6(c.get_covariances(which)), _cov6(g.get_covs(which))
        d = np.abs(a6 - b6).max(1)  # PLANE: eigenvalues (1e-3, 1, 1) -> entries of order 1
        print("%s %s RBF covariances vs cuda-compat oracle: max |d| %.2e, 99.9th percentile %.2e, exactly equal %.1f %%" % (name, which, d.max(), np.percentile(d, 99.9), 100.0 * np.mean(d == 0)))
        # the same float sums in the same order; what differs: expf (correctly rounded here, glibc's there: a last-ulp difference in ~0.1 % of the
        # weights) and atan2f / cosf / sinf of the closed-form eigen solver, amplified by 1 / (eigenvalue gap) on near-degenerate neighbourhoods
        assert np.percentile(d, 99) < 1e-5 and np.mean(d < 1e-4) > 0.999, (d.max(), np.percentile(d, 99))
        worst[which] = d.max()
    # the voxel map: same voxels, same counts; means BIT-EQUAL (the same float sums of the same points in the same order); a voxel's covariance is the
    # float mean of its points' covariances, so it differs by at most what its worst point does (a near-degenerate neighbourhood whose eigenvectors
    # turn on a last-ulp difference of the weights: one point in 100k moves by 0.1) and, over all voxels, as little as the points do
    oc, on, om, ov = g.get_voxelmap()
    ec, en, em, ev = c.get_voxelmap()
    assert util.voxel_dict(ec, en) == util.voxel_dict(oc, on)
    od, ed = util.voxel_dict(oc, om, _cov6(ov)), util.voxel_dict(ec, em, _cov6(ev))
    dm = max(np.abs(np.asarray(ed[k][0], np.float64) - od[k][0]).max() for k in od)
    dcv = np.array([np.abs(ed[k][1] - od[k][1]).max() for k in od])
    print("%s voxel means max |d| %.2e, voxel covariances max |d| %.2e, 99.9th percentile %.2e over %d voxels" % (name, dm, dcv.max(), np.percentile(dcv, 99.9), len(od)))
    assert dm == 0.0, dm
    assert dcv.max() <= worst["target"] + 1e-6 and np.percentile(dcv, 99) < 1e-5, (dcv.max(), worst["target"], np.percentile(dcv, 99))
    T0 = np.eye(4)
    eo, Ho, bo = g.linearize(T0)
    e, H, b = c.linearize(T0)
    util.assert_same_correspondences(c, g, ordered=True)
    assert abs(e - eo) < 1e-4 * abs(eo) and util.rel_err(H, Ho) < 1e-4, (abs(e - eo) / abs(eo), util.rel_err(H, Ho))
    ro, r = g.align(), c.align()
    fo = g.fitness()
    f = c.fitness_score(r["T"].astype(np.float32).astype(np.float64))
    assert r["converged"] and ro["converged"]
    assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"], (r, ro)
    util.assert_same_correspondences(c, g, ordered=True)
    dT = util.rel_err(r["T"], ro["T"])
    print("%s CUDA_COMPAT RBF end to end: pose rel %.2e, fitness %.6f vs %.6f, %d linearisations" % (name, dT, f, fo, r["num_linearize"]))
    assert dT < 1e-4, dT
    assert abs(f - fo) < 1e-4 * fo, (f, fo)
    c.close()
    return r


@pytest.mark.parametrize("search", ["DIRECT1", "DIRECT27"])
def test_vgicp_cuda_compat_rbf_mode_end_to_end_on_the_bundled_pair(O, search):
    """VERDICT r5 Missing #3(a): FastVGICPCuda's RBF estimator (covariance_estimation_rbf.cu:40-109,120-150: float sums per block of 512
    candidates in index order, blocks added in order, uncentred finalize) + the float voxel sums (gaussian_voxelmap.cu:164-171) in
    FVH_COMPUTE_CUDA_COMPAT, END TO END against oracle/cuda_compat.cpp's cov_mode 1: covariances to float rounding, identical
    correspondence lists, pose and fitness to north_star's 1e-4 with equal iteration counts."""
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    cs, os_ = {"DIRECT1": (capi.DIRECT1, O.DIRECT1), "DIRECT27": (capi.DIRECT27, O.DIRECT27)}[search]
    r = _check_rbf_compat(O, tgt, src, cs, os_, 1.0, "bundled " + search)
    # and the mode is what separates the two arithmetics: the fp64 engine with its own (centred, culled, fp64-reduced) RBF sweep differs by 1e-3 .. 3e-3
    # of the pose (the uncentred float sums over ~100 weighted neighbours at 50 m range; the oracle leg alone moves by up to 2e-4 when its input is
    # merely reordered, tests/test_oracle.py::test_cuda_compat_order_spread)
    c64 = _rbf_engine(tgt, src, cs, capi.COMPUTE_FP64)
    d64 = util.rel_err(r["T"], c64.align()["T"])
    print("bundled %s: CUDA_COMPAT RBF vs the fp64 engine's RBF: pose rel %.2e" % (search, d64))
    assert 0 < d64 < 1e-2, d64
    c64.close()


def test_vgicp_cuda_compat_rbf_mode_c3_100k(O):
    """BASELINE configs[2] (synthetic 100k <-> 100k, voxel resolution 0.5, RBF covariances) in the CUDA class's arithmetic."""
    from fast_gicp_amd import capi
    tgt, src, _ = util.synthetic_pair(100000, 100000)
    _check_rbf_compat(O, tgt, src, capi.DIRECT1, O.DIRECT1, 0.5, "C3 100k DIRECT1")


@pytest.mark.parametrize("mode", ["D2D", "P2D"])
def test_ndt_cuda_compat_mode_end_to_end(O, mode):
    """VERDICT r5 Missing #3(b, c): NDTCuda in FVH_COMPUTE_CUDA_COMPAT -- float voxel coordinates (vector3_hash.cuh:35-38), float UNCENTRED voxel
    sums in point order (gaussian_voxelmap.cu:122-148,178-198), Eigen's closed-form float eigen solver for MIN_EIG (ndt_cuda.cu:128,139), float
    cost terms -- against the float restatement at north_star's 1e-4 (the fp32 mode above, whose voxel sums are fp64, is held to 1e-3: the
    uncentred float sums ARE the difference). The oracle leg itself moves by 1e-5 .. 6e-5 when its input is reordered
    (tests/test_oracle.py::test_cuda_compat_order_spread): both sides take index order."""
    from fast_gicp_amd import capi, workloads
    om, cm = {"D2D": (O.D2D, capi.NDT_D2D), "P2D": (O.P2D, capi.NDT_P2D)}[mode]
    t, s = util.bundled_pair(origin_filter=False, leaf=0.2, exact_voxelgrid=True)
    f0 = O.approx_voxelgrid(workloads.lidar_frame(2), 0.25)
    f1 = O.approx_voxelgrid(workloads.lidar_frame(3), 0.25)
    for name, tgt, src in (("gicp_test pair", t, s), ("lidar frames", f0, f1)):
        g = O.CudaCompatNDT(mode=om, search=O.DIRECT7)
        g.set_target(tgt); g.set_source(src)
        ro = g.align()
        d = capi.NDTCore(0)
        d.set_precision(capi.COMPUTE_CUDA_COMPAT)
        d.set_distance_mode(cm); d.set_neighbor_search_method(capi.DIRECT7); d.set_resolution(1.0)
        d.set_target_cloud(tgt); d.set_source_cloud(src)
        r = d.align()
        for which in (("target", "source") if mode == "D2D" else ("target",)):
            oc, on, omn, ov = g.get_voxelmap(which)
            ec, en, em, ev = d.get_voxelmap(which)
            assert util.voxel_dict(ec, en) == util.voxel_dict(oc, on), (name, which)
            od, ed = util.voxel_dict(oc, omn, _cov6(ov)), util.voxel_dict(ec, em, _cov6(ev))
            dm = max(np.abs(np.asarray(ed[k][0], np.float64) - od[k][0]).max() for k in od)
            dc = np.array([np.abs(ed[k][1] - od[k][1]).max() / max(np.abs(od[k][1]).max(), 1e-30) for k in od])
            print("NDT %s, %s, %s voxels: means max |d| %.2e; covariances rel: max %.2e, 99th percentile %.2e, exactly equal %.1f %%" % (mode, name, which, dm, dc.max(), np.percentile(dc, 99), 100.0 * np.mean(dc == 0)))
            assert dm == 0.0, dm  # the same float sums in the same order, the same float division
            assert np.percentile(dc, 99) < 1e-4, np.percentile(dc, 99)  # (device atan2f / cosf / sinf against glibc's inside the eigen solver)
        assert r["converged"] and ro["converged"], name
        assert r["num_linearize"] == ro["num_linearize"] and r["num_error_evals"] == ro["num_error_evals"], (name, r, ro)
        dT = util.rel_err(r["T"], ro["T"])
        fit, fo = d.fitness_score(r["T"].astype(np.float32).astype(np.float64)), g.fitness()
        print("NDT %s, %s: CUDA_COMPAT engine vs cuda-compat oracle: pose rel %.2e, fitness %.6f vs %.6f" % (mode, name, dT, fit, fo))
        assert dT < 1e-4, (name, dT)
        assert abs(fit - fo) < 1e-4 * fo, (name, fit, fo)
        util.assert_same_correspondences(d, g, d2d=(mode == "D2D"), ordered=True)
        d.close()


def test_vgicp_cuda_compat_voxel_sums_on_injected_covariances(O):
    """gaussian_voxelmap.cu:164-171 on its own: fed the oracle leg's point covariances, the engine's CUDA_COMPAT voxel records are the oracle's
    float sums BIT FOR BIT (means, and the six covariance entries the engine stores)."""
    from fast_gicp_amd import capi
    tgt, src = util.bundled_pair()
    g = O.CudaCompatVGICP(search=O.DIRECT1)
    g.set_target(tgt); g.set_source(src); g.prepare()
    c = capi.VGICPCore(0)
    c.set_precision(capi.COMPUTE_CUDA_COMPAT)
    c.set_target_cloud(tgt); c.set_target_covariances(g.get_covs("target")); c.create_target_voxelmap()
    oc, on, om, ov = g.get_voxelmap()
    ec, en, em, ev = c.get_voxelmap()
    od, ed = util.voxel_dict(oc, on, om.astype(np.float32), _cov6(ov).astype(np.float32)), util.voxel_dict(ec, en, em.astype(np.float32), _cov6(ev).astype(np.float32))
    assert set(od) == set(ed)
    for k in od:
        assert od[k][0] == ed[k][0]
        assert np.array_equal(od[k][1], ed[k][1]), (k, od[k][1], ed[k][1])
        assert np.array_equal(od[k][2], ed[k][2]), (k, od[k][2], ed[k][2])
    c.close()


@pytest.mark.parametrize("resolution", [0.3, 0.7])
def test_cuda_compat_voxel_coordinates_are_float_at_a_resolution_that_is_not_a_power_of_two(O, resolution):
    """vector3_hash.cuh:35-38 computes the voxel coordinate in FLOAT: (x / resolution - 0.5).floor(). At resolutions 1.0 / 0.5 that equals the fp64
    coordinate of the CPU class for every float x; at 0.3 or 0.7 points within float rounding of a voxel face land in the neighbouring voxel.
    FVH_COMPUTE_CUDA_COMPAT takes the float coordinate in the map build AND in the lookups: voxel sets, counts and the correspondence list equal the
    cuda-compat oracle's exactly; the fp64 engine on the same data differs from them in a few voxels (the test is not vacuous)."""
    from fast_gicp_amd import capi
    rng = np.random.default_rng(5)
    tgt, src = util.bundled_pair()
    # many points EXACTLY on voxel faces in fp64 terms (multiples of the resolution plus half a voxel), where float and double division disagree most often
    faces = (np.round(rng.uniform(-60, 60, size=(4000, 3)) / resolution) * resolution + 0.5 * resolution).astype(np.float32)
    tgt = np.vstack([tgt, faces]).astype(np.float32)
    src = np.vstack([src, faces[::2] + np.float32(1e-3)]).astype(np.float32)
    g = O.CudaCompatVGICP(search=O.DIRECT7, resolution=resolution)
    g.set_target(tgt); g.set_source(src); g.prepare()

    def engine(precision):
        c = capi.VGICPCore(0)
        c.set_precision(precision); c.set_resolution(resolution); c.set_neighbor_search_method(capi.DIRECT7)
        c.set_target_cloud(tgt); c.set_source_cloud(src)
        c.set_target_covariances(g.get_covs("target")); c.set_source_covariances(g.get_covs("source"))
        c.create_target_voxelmap()
        return c
    c = engine(capi.COMPUTE_CUDA_COMPAT)
    oc, on, om, _ = g.get_voxelmap()
    ec, en, em, _ = c.get_voxelmap()
    assert util.voxel_dict(ec, en) == util.voxel_dict(oc, on)
    od, ed = util.voxel_dict(oc, om.astype(np.float32)), util.voxel_dict(ec, em.astype(np.float32))
    assert all(np.array_equal(od[k][0], ed[k][0]) for k in od)  # float sums in point order: the means bit for bit
    T = util.random_pose(np.random.default_rng(9), max_angle_deg=1.0, max_trans=0.3)
    g.linearize(T); c.linearize(T)
    util.assert_same_correspondences(c, g, ordered=True)
    c64 = engine(capi.COMPUTE_FP64)
    dc, dn, _, _ = c64.get_voxelmap()
    assert util.voxel_dict(dc, dn) != util.voxel_dict(oc, on), "the fp64 coordinate gave the same voxels: this data does not exercise the difference"
    c.close(); c64.close()
