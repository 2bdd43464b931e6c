"""Shared test helpers: bundled-pair preprocessing (via the ORACLE restatement of align.cpp:118-147),
synthetic LiDAR-like scenes, comparison helpers."""
import functools
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")


@functools.lru_cache(maxsize=None)
def bundled_pair(origin_filter=True, leaf=0.1, exact_voxelgrid=False):
    """(target, source) float32 Nx3. Default = HEAD align.cpp preprocessing -> 17,047 / 17,334 points."""
    from oracle import oracle as O
    t = O.load_pcd(os.path.join(DATA, "251370668.pcd"))
    s = O.load_pcd(os.path.join(DATA, "251371071.pcd"))
    if origin_filter:
        t, s = O.remove_origin(t), O.remove_origin(s)
    f = O.voxelgrid if exact_voxelgrid else O.approx_voxelgrid
    return f(t, leaf), f(s, leaf)


def relative_pose():
    return np.loadtxt(os.path.join(DATA, "relative.txt"))


from fast_gicp_amd.workloads import pose_error, random_pose, synthetic_scene, synthetic_pair, lidar_pose, lidar_frame  # noqa: E402,F401  (generators live in the package: bench.py uses them too)


def voxel_dict(coords, *arrays):
    return {tuple(c): tuple(a[i] for a in arrays) for i, c in enumerate(np.asarray(coords))}


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)




def cov_error_bound(got, ref, raw, input_rel, gaps="01", storage_rel=1.2e-7, factor=8.0):
    """Error of eigen-regularised covariances (PLANE, MIN_EIG, ...) against a conditioning-aware bound, per point.

    The regularised matrix is a function of the eigenvectors of the raw covariance `raw`; a relative perturbation
    `input_rel` of the raw matrix (fp32 sums of the RBF kernel, 1e-13 for the fp64 k-NN path) turns the eigenvectors by
    about input_rel * lambda_max / gap, where gap is the distance of the eigenvalue(s) the method depends on
    (PLANE: only the smallest -> gaps="01"; the others: the smaller of the two gaps -> gaps="min").  So
        |got - ref|_max  <=  storage_rel * scale  +  factor * input_rel * lambda_max / gap * scale
    with scale = |ref|_max of that point and storage_rel the fp32 output rounding.  Points where the second term says
    nothing (>= 0.5 * scale: eigenvectors undefined at this input precision) are returned as `degenerate` so that the
    caller can bound their NUMBER; every other point is checked against its bound -- no quantiles.
    Returns (err, bound, degenerate)."""
    got, ref, raw = (np.asarray(a, np.float64) for a in (got, ref, raw))
    w = np.linalg.eigvalsh(raw)
    lam = np.maximum(np.abs(w).max(axis=1), 1e-300)
    gap = (w[:, 1] - w[:, 0]) if gaps == "01" else np.minimum(w[:, 1] - w[:, 0], w[:, 2] - w[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        amp = np.where(gap > 1e-14 * lam, factor * input_rel * lam / gap, np.inf)  # zero matrix (isolated point) / exact tie: undefined
    scale = np.abs(ref).max(axis=(1, 2))
    err = np.abs(got - ref).max(axis=(1, 2))
    bound = (storage_rel + amp) * scale
    return err, bound, amp >= 0.5


def sums_close(e, H, b, eo, Ho, bo, tol):
    """err / H / b of one evaluation against a reference at relative tolerance `tol`. b = sum w J^T M e is a sum of
    CANCELLING terms near the optimum (it is the gradient), so it is held to tol x its Cauchy-Schwarz scale
    sqrt(H_ii * err) >= |b_i| rather than to tol x |b|: an input rounding of relative size tol moves every term by that
    much, not the (arbitrarily small) total."""
    H, Ho, b, bo = (np.asarray(a, np.float64) for a in (H, Ho, b, bo))
    ok_e = abs(e - eo) <= tol * abs(eo)
    ok_H = rel_err(H, Ho) <= tol
    scale_b = np.sqrt(np.abs(np.diag(Ho)) * abs(eo))
    ok_b = bool(np.all(np.abs(b - bo) <= tol * np.maximum(scale_b, 1e-300)))
    return ok_e and ok_H and ok_b


def engine_corr_rows(c, d2d=False):
    """The engine's correspondence list (fvh_*_get_voxel_correspondences: (source element, voxel index in getter order), offset-major)
    resolved to rows {source element, source voxel x y z, target voxel x y z} -- voxels by COORDINATE, so the rows do not depend on
    anybody's voxel numbering. VGICP / NDT P2D: the source element is the point index (source voxel columns 0); NDT D2D: the source
    element IS a source voxel, numbered differently by the engine and the oracle -> column 0 is zeroed, its coordinate identifies it."""
    from fast_gicp_amd import capi
    pairs = np.asarray(c.get_voxel_correspondences(), np.int64).reshape(-1, 2)
    rows = np.zeros((len(pairs), 7), np.int64)
    if isinstance(c, capi.NDTCore):
        tcoords = np.asarray(c.get_voxelmap("target")[0], np.int64)
        if d2d:
            rows[:, 1:4] = np.asarray(c.get_voxelmap("source")[0], np.int64)[pairs[:, 0]]
        else:
            rows[:, 0] = pairs[:, 0]
    else:
        tcoords = np.asarray(c.get_voxelmap()[0], np.int64)
        rows[:, 0] = pairs[:, 0]
    rows[:, 4:7] = tcoords[pairs[:, 1]]
    return rows


def oracle_corr_rows(g, d2d=False):
    rows = np.asarray(g.correspondences(), np.int64)
    if d2d:
        rows = rows.copy()
        rows[:, 0] = 0
    return rows


def sort_rows(rows):
    rows = np.asarray(rows)
    return rows[np.lexsort(rows.T[::-1])] if len(rows) else rows


def assert_same_correspondences(c, g, d2d=False, ordered=False):
    """Index-level parity: the engine's (source element, voxel) pairs EQUAL the oracle's -- as a set, or (ordered=True: the cuda-compat
    leg, whose list is offset-major like find_voxel_correspondences.cu:84-111 and like the engine's getter) as a list."""
    got, ref = engine_corr_rows(c, d2d), oracle_corr_rows(g, d2d)
    assert len(got) == len(ref), (len(got), len(ref))
    assert len(np.unique(got, axis=0)) == len(got), "duplicate pairs in the engine's list"
    if ordered and not d2d:
        assert np.array_equal(got, ref), "correspondence lists differ (order included)"
    else:
        assert np.array_equal(sort_rows(got), sort_rows(ref)), "correspondence pair sets differ"
