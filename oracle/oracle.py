"""ORACLE (test infrastructure, NOT product code).

ctypes front-end of oracle/liboracle.so -- the fp64 CPU restatement of the reference's
FastVGICP / NDT hot path (see oracle/vgicp_oracle.hpp for what it follows and how it is
pinned).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (fast_gicp_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# enums (reference gicp_settings.hpp:7-11, ndt_settings.hpp:6)
NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS = range(5)
DIRECT27, DIRECT7, DIRECT1, DIRECT_RADIUS = range(4)
P2D, D2D = 0, 1


class Result(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("H", C.c_double * 36), ("converged", C.c_int), ("nr_iterations", C.c_int),
                ("num_linearize", C.c_int), ("num_error_evals", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("vgicp_oracle.cpp", "oracle_capi.cpp", "cuda_compat.cpp", "vgicp_oracle.hpp", "oracle_math.hpp", "oracle_align.cpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dbl = C.c_double
        vp = C.c_void_p
        L.orc_fitness.restype = dbl
        L.orc_vgicp_create.restype = vp
        L.orc_ndt_create.restype = vp
        L.orc_ccv_create.restype = vp
        L.orc_ccn_create.restype = vp
        for f in ("orc_vgicp_cuda_compat_sums", "orc_vgicp_linearize", "orc_vgicp_compute_error", "orc_vgicp_fitness", "orc_vgicp_bench", "orc_ndt_linearize", "orc_ndt_compute_error", "orc_ndt_fitness",
                  "orc_ccv_linearize", "orc_ccv_compute_error", "orc_ccv_fitness", "orc_ccn_linearize", "orc_ccn_compute_error", "orc_ccn_fitness"):
            getattr(L, f).restype = dbl
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load_pcd(path):
    n = lib().orc_load_pcd(path.encode(), None, 0)
    if n < 0:
        raise IOError("cannot read " + path)
    out = np.empty((n, 3), np.float32)
    lib().orc_load_pcd(path.encode(), _p(out), n)
    return out


def remove_origin(xyz):
    a = _f32(xyz).copy()
    n = lib().orc_remove_origin(_p(a), len(a))
    return a[:n].copy()


def approx_voxelgrid(xyz, leaf):
    a = _f32(xyz)
    out = np.empty_like(a)
    n = lib().orc_approx_voxelgrid(_p(a), len(a), C.c_float(leaf), _p(out))
    return out[:n].copy()


def voxelgrid(xyz, leaf):
    a = _f32(xyz)
    out = np.empty_like(a)
    n = lib().orc_voxelgrid(_p(a), len(a), C.c_float(leaf), _p(out))
    return out[:n].copy()


def knn(xyz, k, threads=0):
    a = _f32(xyz)
    idx = np.empty((len(a), k), np.int32)
    lib().orc_knn(_p(a), len(a), k, threads or os.cpu_count(), _p(idx))
    return idx


def knn_query(cloud, queries, k):
    a, q = _f32(cloud), _f32(queries)
    idx = np.empty((len(q), k), np.int32)
    sq = np.empty((len(q), k), np.float32)
    lib().orc_knn_query(_p(a), len(a), _p(q), len(q), k, _p(idx), _p(sq))
    return idx, sq


def covariances_knn(xyz, k=20, reg=PLANE, idx=None, threads=0):
    a = _f32(xyz)
    out = np.empty((len(a), 3, 3), np.float64)
    ip = None
    if idx is not None:
        idx = np.ascontiguousarray(idx, np.int32)
        ip = _p(idx)
    lib().orc_covariances_knn(_p(a), len(a), k, ip, reg, threads or os.cpu_count(), _p(out))
    return out


def covariances_rbf(xyz, kernel_width, max_dist, reg=PLANE, threads=0):
    a = _f32(xyz)
    out = np.empty((len(a), 3, 3), np.float64)
    lib().orc_covariances_rbf(_p(a), len(a), C.c_double(kernel_width), C.c_double(max_dist), reg, threads or os.cpu_count(), _p(out))
    return out


def regularize(cov, reg):
    a = _f64(cov).reshape(3, 3)
    out = np.empty((3, 3), np.float64)
    lib().orc_regularize(_p(a), reg, _p(out))
    return out


def _voxel_out(n):
    return (np.empty((n, 3), np.int32), np.empty(n, np.int32), np.empty((n, 3), np.float64), np.empty((n, 3, 3), np.float64))


ADDITIVE, ADDITIVE_WEIGHTED, MULTIPLICATIVE = range(3)  # VoxelAccumulationMode (gicp_settings.hpp:10)


def voxelmap_vgicp(xyz, covs, res, mode=ADDITIVE):
    a, c = _f32(xyz), _f64(covs)
    coords, num, means, vc = _voxel_out(len(a))
    nv = lib().orc_voxelmap_vgicp_mode(_p(a), _p(c), len(a), C.c_double(res), int(mode), _p(coords), _p(num), _p(means), _p(vc))
    return coords[:nv].copy(), num[:nv].copy(), means[:nv].copy(), vc[:nv].copy()


def voxelmap_ndt(xyz, res):
    a = _f32(xyz)
    coords, num, means, vc = _voxel_out(len(a))
    nv = lib().orc_voxelmap_ndt(_p(a), len(a), C.c_double(res), _p(coords), _p(num), _p(means), _p(vc))
    return coords[:nv].copy(), num[:nv].copy(), means[:nv].copy(), vc[:nv].copy()


def neighbor_offsets(method, radius=0.0):
    out = np.empty((4096, 3), np.int32)
    n = lib().orc_neighbor_offsets(method, C.c_double(radius), _p(out), 4096)
    return out[:n].copy()


def se3_exp(a):
    a = _f64(a)
    T = np.empty((4, 4), np.float64)
    lib().orc_se3_exp(_p(a), _p(T))
    return T


def fitness(source, target, T):
    s, t, T = _f32(source), _f32(target), _f64(T)
    return lib().orc_fitness(_p(s), len(s), _p(t), len(t), _p(T))


class _Reg:
    _prefix = ""

    def _call(self, name, *args):
        return getattr(lib(), self._prefix + name)(C.c_void_p(self.h), *args)

    def __del__(self):
        if getattr(self, "h", None):
            self._call("destroy")
            self.h = None

    def set_lm(self, max_iterations=64, rotation_epsilon=2e-3, transformation_epsilon=5e-4, lm_max_iterations=10, init_lambda_factor=1e-9, debug=False):
        args = [max_iterations, C.c_double(rotation_epsilon), C.c_double(transformation_epsilon), lm_max_iterations, C.c_double(init_lambda_factor)]
        if self._prefix == "orc_vgicp_":
            args.append(int(debug))
        self._call("set_lm", *args)

    def set_optimizer(self, name):
        """lsq_optimizer_type_: "LM" (default, lsq_registration_impl.hpp:15) or "GN" (step_gn, :108-121)."""
        self._call("set_optimizer", 1 if name == "GN" else 0)

    def set_target(self, xyz):
        a = _f32(xyz)
        self._call("set_target", _p(a), len(a))
        self.nt = len(a)

    def set_source(self, xyz):
        a = _f32(xyz)
        self._call("set_source", _p(a), len(a))
        self.ns = len(a)

    def swap(self):
        self._call("swap")
        self.ns, self.nt = getattr(self, "nt", 0), getattr(self, "ns", 0)

    def prepare(self):
        self._call("prepare")

    def linearize(self, T):
        T = _f64(T)
        H = np.empty((6, 6), np.float64)
        b = np.empty(6, np.float64)
        e = self._call("linearize", _p(T), _p(H), _p(b))
        return e, H, b

    def compute_error(self, T):
        T = _f64(T)
        return self._call("compute_error", _p(T))

    def num_correspondences(self):
        return self._call("num_correspondences")

    def correspondences(self):
        """Every correspondence of the last linearize() as a row {source element, source voxel x y z (NDT D2D, else 0), target voxel x y z
        (FastGICP mode: target point index, 0, 0)}, in the oracle's list order (index-level parity tests compare them as sets)."""
        out = np.empty((max(self.num_correspondences(), 1), 7), np.int32)
        n = self._call("get_correspondences", _p(out))
        return out[:n].copy()

    def align(self, guess=None):
        g = _f64(np.eye(4) if guess is None else guess)
        r = Result()
        self._call("align", _p(g), C.byref(r))
        return dict(T=np.array(r.T).reshape(4, 4), H=np.array(r.H).reshape(6, 6), converged=bool(r.converged), iterations=r.nr_iterations + 1,
                    num_linearize=r.num_linearize, num_error_evals=r.num_error_evals)

    def fitness(self):
        return self._call("fitness")


class FastVGICP(_Reg):
    """fp64 restatement of the reference CPU FastVGICP (fast_vgicp_impl.hpp)."""
    _prefix = "orc_vgicp_"

    def __init__(self, threads=0, k=20, reg=PLANE, resolution=1.0, search=DIRECT1, cov_mode=0, kernel_width=0.5, kernel_max_dist=3.0, round_fp32=False):
        self.h = lib().orc_vgicp_create()
        self._call("set_params", threads, k, reg, C.c_double(resolution), search, cov_mode, C.c_double(kernel_width), C.c_double(kernel_max_dist), int(round_fp32))

    def set_gicp_mode(self, on=True, max_correspondence_distance=3.4028234663852886e38):
        """FastGICP (fast_gicp_impl.hpp:118-240): nearest-target-point correspondences instead of voxels."""
        self._call("set_gicp_mode", int(on), C.c_double(max_correspondence_distance))

    def set_voxel_accumulation_mode(self, mode):
        """FastVGICP::setVoxelAccumulationMode (fast_vgicp_impl.hpp:41-43)."""
        self._call("set_voxel_mode", int(mode))

    def cuda_compat_sums(self, T, derivatives=True):
        """compute_derivatives.cu:50-103 arithmetic in float over the correspondences / linearisation pose of the last linearize()."""
        T = _f64(T)
        if not derivatives:
            return self._call("cuda_compat_sums", _p(T), None, None)
        H = np.empty((6, 6), np.float64)
        b = np.empty(6, np.float64)
        e = self._call("cuda_compat_sums", _p(T), _p(H), _p(b))
        return e, H, b

    def set_target_covs(self, covs):
        c = _f64(covs)
        self._call("set_target_covs", _p(c))

    def set_source_covs(self, covs):
        c = _f64(covs)
        self._call("set_source_covs", _p(c))

    def clear_source(self):
        self._call("clear_source")

    def clear_target(self):
        self._call("clear_target")

    def get_covs(self, which):
        n = self.nt if which == "target" else self.ns
        out = np.empty((n, 3, 3), np.float64)
        self._call("get_covs", 1 if which == "target" else 0, _p(out))
        return out

    def set_voxelmap(self, coords, num, means, covs):
        """Test leg: the target voxel map becomes exactly these records (e.g. the engine's own fp32-stored ones)."""
        c, n, m, v = np.ascontiguousarray(coords, np.int32), np.ascontiguousarray(num, np.int32), _f64(means), _f64(covs)
        self._call("set_voxelmap", len(n), _p(c), _p(n), _p(m), _p(v))

    def get_voxelmap(self):
        coords, num, means, vc = _voxel_out(self.nt)
        nv = self._call("get_voxelmap", _p(coords), _p(num), _p(means), _p(vc))
        return coords[:nv].copy(), num[:nv].copy(), means[:nv].copy(), vc[:nv].copy()

    def bench(self, target, source, mode, loops=100):
        """mode 0 single / 1 N-times / 2 N-times-reuse (src/align.cpp:51-104). Returns (ms, fitness)."""
        t, s = _f32(target), _f32(source)
        fit = C.c_double(0)
        ms = self._call("bench", _p(t), len(t), _p(s), len(s), mode, loops, C.byref(fit))
        return ms, fit.value


class NDT(_Reg):
    """fp64 restatement of NDTCuda's formulas (ndt_cuda.cu, ndt_compute_derivatives.cu)."""
    _prefix = "orc_ndt_"

    def __init__(self, threads=0, resolution=1.0, mode=D2D, search=DIRECT7, radius=0.0, round_fp32=False):
        self.h = lib().orc_ndt_create()
        self._call("set_params", threads, C.c_double(resolution), mode, search, C.c_double(radius))
        if round_fp32:
            self._call("set_round_fp32", 1)

    def set_voxelmap(self, which, coords, num, means, covs):
        """Test leg: a voxel map becomes exactly these records (source order = the D2D iteration order)."""
        c, n, m, v = np.ascontiguousarray(coords, np.int32), np.ascontiguousarray(num, np.int32), _f64(means), _f64(covs)
        self._call("set_voxelmap", 1 if which == "target" else 0, len(n), _p(c), _p(n), _p(m), _p(v))

    def get_voxelmap(self, which):
        n = self.nt if which == "target" else self.ns
        coords, num, means, vc = _voxel_out(n)
        nv = self._call("get_voxelmap", 1 if which == "target" else 0, _p(coords), _p(num), _p(means), _p(vc))
        return coords[:nv].copy(), num[:nv].copy(), means[:nv].copy(), vc[:nv].copy()


# ---------------------------------------------------------------------------------------------------
# "cuda-compat" leg (oracle/cuda_compat.cpp): fp32 restatement of the reference's DEVICE path end to end
# ---------------------------------------------------------------------------------------------------
class _Compat(_Reg):
    def set_lm(self, max_iterations=64, rotation_epsilon=2e-3, transformation_epsilon=5e-4, lm_max_iterations=10, init_lambda_factor=1e-9):
        self._call("set_lm", max_iterations, C.c_double(rotation_epsilon), C.c_double(transformation_epsilon), lm_max_iterations, C.c_double(init_lambda_factor))

    def corr_history(self):
        """Correspondence-list length after every update_correspondences() of the last align()."""
        out = np.empty(4096, np.int32)
        n = self._call("corr_history", _p(out), 4096)
        return out[:n].tolist()


class CudaCompatVGICP(_Compat):
    """FastVGICPCuda + FastVGICPCudaCore in float (covariance_estimation.cu, covariance_regularization.cu, gaussian_voxelmap.cu,
    find_voxel_correspondences.cu, compute_derivatives.cu); k = 20 always, like the reference's CUDA class."""
    _prefix = "orc_ccv_"

    def __init__(self, threads=0, reg=PLANE, resolution=1.0, search=DIRECT1, radius=0.0, cov_mode=0, kernel_width=0.5, kernel_max_dist=3.0):
        self.h = lib().orc_ccv_create()
        self._call("set_params", threads, reg, C.c_double(resolution), search, C.c_double(radius), cov_mode, C.c_double(kernel_width), C.c_double(kernel_max_dist))

    def get_covs(self, which):
        n = self.nt if which == "target" else self.ns
        out = np.empty((n, 3, 3), np.float64)
        self._call("get_covs", 1 if which == "target" else 0, _p(out))
        return out

    def get_voxelmap(self):
        coords, num, means, vc = _voxel_out(self.nt)
        nv = self._call("get_voxelmap", _p(coords), _p(num), _p(means), _p(vc))
        return coords[:nv].copy(), num[:nv].copy(), means[:nv].copy(), vc[:nv].copy()


class CudaCompatNDT(_Compat):
    """NDTCuda + NDTCudaCore in float (gaussian_voxelmap.cu NDT branch + MIN_EIG, ndt_compute_derivatives.cu)."""
    _prefix = "orc_ccn_"

    def __init__(self, threads=0, resolution=1.0, mode=D2D, search=DIRECT7, radius=0.0):
        self.h = lib().orc_ccn_create()
        self._call("set_params", threads, C.c_double(resolution), mode, search, C.c_double(radius))

    def get_voxelmap(self, which):
        n = self.nt if which == "target" else self.ns
        coords, num, means, vc = _voxel_out(n)
        nv = self._call("get_voxelmap", 1 if which == "target" else 0, _p(coords), _p(num), _p(means), _p(vc))
        return coords[:nv].copy(), num[:nv].copy(), means[:nv].copy(), vc[:nv].copy()


def cc_eig3(A):
    """Eigen::SelfAdjointEigenSolver<Matrix3f>::computeDirect as restated in cuda_compat.cpp: (values ascending, vectors in columns)."""
    a = _f64(A).reshape(3, 3)
    w, V = np.empty(3, np.float64), np.empty((3, 3), np.float64)
    lib().orc_cc_eig3(_p(a), _p(w), _p(V))
    return w, V


def cc_regularize(A, reg):
    a = _f64(A).reshape(3, 3)
    out = np.empty((3, 3), np.float64)
    lib().orc_cc_regularize(_p(a), reg, _p(out))
    return out
