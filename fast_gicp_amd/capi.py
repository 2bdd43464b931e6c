"""ctypes binding of the C ABI in include/fast_vgicp_hip.h (libfast_vgicp_hip.so).

This is the only way Python reaches the engine: plain pointers and sizes, no torch types.
There is no CPU fallback -- if the shared library is missing, importing the handles fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

# enum ordinals (== the reference's enum class ordinals, gicp_settings.hpp:7-11 / ndt_settings.hpp:6)
REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = range(5)
DIRECT27, DIRECT7, DIRECT1, DIRECT_RADIUS = range(4)
NDT_P2D, NDT_D2D = 0, 1
COMPUTE_FP64, COMPUTE_FP32, COMPUTE_CUDA_COMPAT = 0, 1, 2
VOXEL_ADDITIVE, VOXEL_ADDITIVE_WEIGHTED, VOXEL_MULTIPLICATIVE = range(3)

EXPORTED_SYMBOLS = None  # filled by _declared_symbols()


class LmParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("rotation_epsilon", C.c_double), ("transformation_epsilon", C.c_double), ("lm_max_iterations", C.c_int),
                ("lm_init_lambda_factor", C.c_double), ("optimizer", C.c_int)]


class LmResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("H", C.c_double * 36), ("final_error", C.c_double), ("converged", C.c_int), ("nr_iterations", C.c_int),
                ("num_linearize", C.c_int), ("num_error_evals", C.c_int), ("lm_failed", C.c_int), ("num_launches", C.c_int)]


class EngineParams(C.Structure):
    """fvh_engine_params (include/fast_vgicp_hip.h): the routes / thresholds / watchdogs of ONE handle"""
    _fields_ = [("struct_size", C.c_int), ("sort_mode", C.c_int), ("sort_items", C.c_int), ("sort_fused_bits", C.c_int), ("sort_two_pass_max", C.c_int),
                ("sort_coop_watchdog_ticks", C.c_ulonglong), ("knn_nearest_first_max_points", C.c_int), ("knn_block", C.c_int), ("coherent_min_points", C.c_int),
                ("bitmap_min_points", C.c_int), ("bitmap_max_bytes", C.c_ulonglong), ("persistent", C.c_int), ("persist_watchdog_ticks", C.c_ulonglong),
                ("peer_watchdog_ticks", C.c_ulonglong), ("lm_everywhere", C.c_int), ("cost_prio", C.c_int), ("cost_split", C.c_int), ("cost_group_max", C.c_int),
                ("cost_max_blocks", C.c_int), ("cost_target_items", C.c_longlong), ("zerocopy_result", C.c_int), ("host_wait_block", C.c_int),
                ("result_query_spins", C.c_ulonglong), ("side_stream", C.c_int), ("pinned_upload_max", C.c_ulonglong), ("zerocopy_upload_max", C.c_ulonglong),
                ("avg_fused", C.c_int), ("nn1_seed", C.c_int)]


def default_engine_params():
    p = EngineParams()
    load().fvh_default_engine_params(C.byref(p))
    return p


class FvhError(RuntimeError):
    pass


_LIB = None


def lib_path():
    return os.environ.get("FVH_LIB_PATH") or _build.LIB_PATH  # override: A/B runs against another build of the same ABI


def load():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise FvhError("libfast_vgicp_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback." % path)
        L = C.CDLL(path)
        L.fvh_vgicp_last_error.restype = C.c_char_p
        L.fvh_ndt_last_error.restype = C.c_char_p
        L.fvh_voxelgrid_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def declared_symbols():
    """Every function declared in include/fast_vgicp_hip.h (parsed from the header)."""
    import re
    hdr = open(_build.HEADER).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fvh_[a-z0-9_]+)\s*\(", hdr)))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return C.c_void_p(a.__array_interface__["data"][0])  # (a.ctypes.data_as costs 3.7 us a call)


def _colmajor16(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T)


def _lm_params(max_iterations=64, rotation_epsilon=2e-3, transformation_epsilon=5e-4, lm_max_iterations=10, lm_init_lambda_factor=1e-9, optimizer=0):
    """optimizer: 0 Levenberg-Marquardt (default), 1 Gauss-Newton (LSQ_OPTIMIZER_TYPE, lsq_registration.hpp:15)"""
    return LmParams(max_iterations, rotation_epsilon, transformation_epsilon, lm_max_iterations, lm_init_lambda_factor, optimizer)


_IDENTITY16 = np.ascontiguousarray(np.eye(4))  # the default guess / parameters of align(): built once (these calls sit between two LM kernels, with the GPU idle)
_DEFAULT_LM = _lm_params()


def _result_dict(r):
    # one view over the struct's 52 leading doubles (the arrays keep `r` alive): this runs between an align and the next launch,
    # with the GPU idle
    a = np.frombuffer(r, dtype=np.float64, count=52)
    return dict(T=a[:16].reshape(4, 4).T, H=a[16:].reshape(6, 6).T, final_error=r.final_error, converged=bool(r.converged),
                nr_iterations=r.nr_iterations, iterations=r.nr_iterations + 1, num_linearize=r.num_linearize, num_error_evals=r.num_error_evals,
                lm_failed=bool(r.lm_failed), num_launches=r.num_launches)


class _Core:
    _prefix = ""

    def __init__(self, device=0):
        self._lib = load()
        if "_fns" not in type(self).__dict__:
            type(self)._fns = {}  # per class (VGICPCore / NDTCore have different prefixes)
        self.device = int(device)
        self.h = C.c_void_p()
        rc = getattr(self._lib, self._prefix + "create")(int(device), C.byref(self.h))
        if rc != 0:
            raise FvhError("%screate failed with status %d" % (self._prefix, rc))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            getattr(self._lib, self._prefix + "destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        fn = self._fns.get(name)
        if fn is None:  # (bound once per class: the attribute lookup on the CDLL sits between an align and the next launch)
            fn = self._fns[name] = getattr(self._lib, self._prefix + name)
        rc = fn(self.h, *args)
        if rc != 0:
            msg = getattr(self._lib, self._prefix + "last_error")(self.h)
            raise FvhError("%s%s: status %d: %s" % (self._prefix, name, rc, msg.decode() if msg else ""))

    # ---- shared API ----
    def get_engine_params(self):
        p = EngineParams()
        self._call("get_engine_params", C.byref(p))
        return p

    def set_engine_params(self, **fields):
        """change some of this handle's engine parameters (names: fvh_engine_params), e.g. set_engine_params(persistent=0, sort_mode=1)"""
        p = self.get_engine_params()
        for k, v in fields.items():
            if k not in dict(EngineParams._fields_) or k == "struct_size":
                raise FvhError("unknown engine parameter %r" % k)
            setattr(p, k, v)
        self._call("set_engine_params", C.byref(p))
        return p

    def set_resolution(self, r):
        self._call("set_resolution", C.c_double(r))

    def set_neighbor_search_method(self, method, radius=-1.0):
        self._call("set_neighbor_search_method", int(method), C.c_double(radius))

    def set_precision(self, p):
        self._call("set_precision", int(p))

    def swap_source_and_target(self):
        self._call("swap_source_and_target")

    def set_source_cloud(self, xyz):
        a = _f32(xyz)
        self._call("set_source_cloud", _p(a), len(a))

    def set_target_cloud(self, xyz):
        a = _f32(xyz)
        self._call("set_target_cloud", _p(a), len(a))

    def set_source_cloud_device(self, ptr, n, stride=3):
        self._call("set_source_cloud_device", C.c_void_p(ptr), int(n), int(stride))

    def set_target_cloud_device(self, ptr, n, stride=3):
        self._call("set_target_cloud_device", C.c_void_p(ptr), int(n), int(stride))

    def update_correspondences(self, T):
        t = _colmajor16(T)
        self._call("update_correspondences", _p(t))

    def compute_error(self, T, derivatives=True):
        t = _colmajor16(T)
        err = C.c_double(0)
        if derivatives:
            H = np.empty((6, 6), np.float64)
            b = np.empty(6, np.float64)
            self._call("compute_error", _p(t), _p(H), _p(b), C.byref(err))
            return err.value, H.T.copy(), b
        self._call("compute_error", _p(t), None, None, C.byref(err))
        return err.value

    def linearize(self, T):
        """LsqRegistration::linearize of the reference wrappers: update_correspondences + compute_error."""
        self.update_correspondences(T)
        return self.compute_error(T, True)

    def align(self, guess=None, **lm):
        g = _IDENTITY16 if guess is None else _colmajor16(guess)
        p = _lm_params(**lm) if lm else _DEFAULT_LM
        r = LmResult()
        self._call("align", _p(g), C.byref(p), C.byref(r))
        return _result_dict(r)

    def fitness_score(self, T, max_range=1.7976931348623157e308):
        t = _colmajor16(T)
        s = C.c_double(0)
        self._call("fitness_score", _p(t), C.c_double(max_range), C.byref(s))
        return s.value

    def synchronize(self):
        self._call("synchronize")

    def set_lm_trace(self, on=True):
        self._call("set_lm_trace", int(on))

    def get_lm_trace(self):
        """Rows {i, y0, yi, rho, lambda, |d|} of the last align (one per trial step), as LsqRegistration's debug print."""
        n = C.c_int(0)
        self._call("get_lm_trace", C.byref(n), None)
        out = np.empty((n.value, 6), np.float64)
        if n.value:
            self._call("get_lm_trace", C.byref(n), _p(out))
        return out

    def profile_enable(self, on=True):
        self._call("profile_enable", int(on))

    def profile_reset(self):
        self._call("profile_reset")

    def profile_get(self, cls):
        ms = C.c_double(0)
        n = C.c_int(0)
        self._call("profile_get", cls.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def comm_init(self, unique_id, nranks, rank):
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._call("comm_init", buf, int(nranks), int(rank))

    def comm_destroy(self):
        self._call("comm_destroy")

    def get_num_correspondences(self):
        n = C.c_int(0)
        self._call("get_num_correspondences", C.byref(n))
        return n.value


def comm_unique_id():
    buf = (C.c_char * 128)()
    rc = load().fvh_comm_unique_id(buf)
    if rc != 0:
        raise FvhError("fvh_comm_unique_id failed: %d" % rc)
    return bytes(buf)


def debug_slot_pool(device=0):
    """(reserved, active, recent) of the process-wide pool that splits a device's co-resident workgroup slots between concurrent aligns."""
    r, a, c = C.c_int(0), C.c_int(0), C.c_int(0)
    load().fvh_debug_slot_pool(int(device), C.byref(r), C.byref(a), C.byref(c))
    return r.value, a.value, c.value


def debug_sort_routes():
    """[cooperative, one workgroup, two-launch passes, four-launch passes]: Morton sorts queued by this process so far, per route"""
    a = (C.c_int * 4)()
    rc = load().fvh_debug_sort_routes(a)
    if rc != 0:
        raise FvhError("fvh_debug_sort_routes: status %d" % rc)
    return list(a)


def debug_xcd_local():
    """(wanted, placement_aborts): whether persistent launches still use XCD-local hand-offs, and how many launches the placement check ended."""
    w, a = C.c_int(0), C.c_int(0)
    load().fvh_debug_xcd_local(C.byref(w), C.byref(a))
    return w.value, a.value


def device_count():
    n = C.c_int(0)
    rc = load().fvh_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class VGICPCore(_Core):
    """fast_gicp::cuda::FastVGICPCudaCore on the HIP engine (same method names, snake_case as in the .cuh)."""
    _prefix = "fvh_vgicp_"

    # ---- pipelined scan streams (include/fast_vgicp_hip.h: fvh_vgicp_align_async ...) ----
    def align_async(self, guess=None, **lm):
        """Launch the LM kernel and return; align_wait() collects the result. In between only prepare_source_device() may be used."""
        self._g = _IDENTITY16 if guess is None else _colmajor16(guess)
        self._lm = _lm_params(**lm) if lm else _DEFAULT_LM
        self._call("align_async", _p(self._g), C.byref(self._lm))

    def align_wait(self):
        r = LmResult()
        self._call("align_wait", C.byref(r))
        return _result_dict(r)

    def prepare_source_device(self, ptr, n, stride=3, k=20, regularization=REG_PLANE, rbf=False, stages=3):
        """The NEXT source cloud (device pointer) into the prepared slot, on the handle's second stream: Morton sort, exact k-NN + covariances
        (or RBF covariances), and the scan's own voxel map -- beside whatever runs on the main stream."""
        self._call("prepare_source_device", C.c_void_p(ptr), int(n), int(stride), int(k), int(regularization), 1 if rbf else 0, int(stages))

    def prepare_source(self, xyz, k=20, regularization=REG_PLANE, rbf=False, stages=2):
        """The same for a host cloud (N x 3 float32): consumed before the call returns."""
        a = np.ascontiguousarray(xyz, np.float32)
        self._call("prepare_source", _p(a), len(a), 3, int(k), int(regularization), 1 if rbf else 0, int(stages))

    def adopt_prepared_source(self):
        self._call("adopt_prepared_source")

    # ---- multi-GPU through peer-mapped exchange regions (no RCCL) ----
    def peer_export(self, max_points):
        """-> (64-byte IPC handle, raw device pointer of the region)"""
        buf = (C.c_char * 64)()
        ptr = C.c_ulonglong(0)
        self._call("peer_export", int(max_points), buf, C.byref(ptr))
        return bytes(buf), ptr.value

    def peer_attach(self, nranks, rank, ranks_on_this_device, ipc_handles, process_local_ptrs=None):
        blob = b"".join(bytes(h) for h in ipc_handles)
        assert len(blob) == 64 * nranks
        hbuf = (C.c_char * len(blob)).from_buffer_copy(blob)
        pbuf = None
        if process_local_ptrs is not None:
            pbuf = (C.c_ulonglong * nranks)(*[int(p) for p in process_local_ptrs])
        self._call("peer_attach", int(nranks), int(rank), int(ranks_on_this_device), hbuf, pbuf)

    def set_target_map_sharding(self, on=True, margin_voxels=2):
        """Multi-GPU: this rank's target voxel map holds the voxels around its tile of the source only (rebuilt per align)."""
        self._call("set_target_map_sharding", 1 if on else 0, int(margin_voxels))

    def debug_spatial_order(self, which="source"):
        """(order, tile_boxes): the Morton order of a cloud (original index per sorted position) and the boxes of its 64-point tiles"""
        n = self.num_points(which)
        order = np.empty(max(n, 1), np.int32)
        boxes = np.empty(((max(n, 1) + 63) // 64, 8), np.float32)
        self._call("debug_get_spatial_order", 0 if which == "source" else 1, _p(order), _p(boxes))
        return order[:n], boxes[: (n + 63) // 64]

    def debug_live_map_voxels(self):
        n = C.c_int(0)
        self._call("debug_get_live_map_voxels", C.byref(n))
        return n.value

    def debug_map_shard(self):
        a, b = C.c_int(0), C.c_int(0)
        self._call("debug_get_map_shard", C.byref(a), C.byref(b))
        return bool(a.value), b.value

    def peer_detach(self):
        self._call("peer_detach")

    def peer_selfcheck(self, timeout_seconds=5.0):
        """Collective: a store of every rank reaches every rank (FvhError with the missing ranks otherwise)."""
        m = C.c_int(0)
        self._call("peer_selfcheck", C.c_double(timeout_seconds), C.byref(m))
        return m.value

    def set_voxel_accumulation_mode(self, mode):
        self._call("set_voxel_accumulation_mode", int(mode))

    def set_kernel_params(self, kernel_width, kernel_max_dist):
        self._call("set_kernel_params", C.c_double(kernel_width), C.c_double(kernel_max_dist))

    def num_points(self, which):
        n = C.c_int(0)
        self._call("get_num_%s_points" % which, C.byref(n))
        return n.value

    def set_source_neighbors(self, k, idx):
        a = np.ascontiguousarray(idx, np.int32)
        self._call("set_source_neighbors", int(k), _p(a))

    def set_target_neighbors(self, k, idx):
        a = np.ascontiguousarray(idx, np.int32)
        self._call("set_target_neighbors", int(k), _p(a))

    def find_source_neighbors(self, k):
        self._call("find_source_neighbors", int(k))

    def find_target_neighbors(self, k):
        self._call("find_target_neighbors", int(k))

    def calculate_source_covariances(self, method=REG_PLANE):
        self._call("calculate_source_covariances", int(method))

    def calculate_target_covariances(self, method=REG_PLANE):
        self._call("calculate_target_covariances", int(method))

    def calculate_source_covariances_rbf(self, method=REG_PLANE):
        self._call("calculate_source_covariances_rbf", int(method))

    def calculate_target_covariances_rbf(self, method=REG_PLANE):
        self._call("calculate_target_covariances_rbf", int(method))

    def set_source_covariances(self, covs):
        a = np.ascontiguousarray(covs, np.float64)
        self._call("set_source_covariances", _p(a))

    def set_target_covariances(self, covs):
        a = np.ascontiguousarray(covs, np.float64)
        self._call("set_target_covariances", _p(a))

    def get_neighbors(self, which):
        k = C.c_int(0)
        self._call("get_%s_neighbors" % which, C.byref(k), None)
        out = np.empty((self.num_points(which), k.value), np.int32)
        self._call("get_%s_neighbors" % which, C.byref(k), _p(out))
        return out

    def get_covariances(self, which):
        out = np.empty((self.num_points(which), 3, 3), np.float32)
        self._call("get_%s_covariances" % which, _p(out))
        return out

    def create_target_voxelmap(self):
        self._call("create_target_voxelmap")

    def get_voxelmap(self):
        n = C.c_int(0)
        self._call("get_num_voxels", C.byref(n))
        nv = n.value
        coords = np.empty((nv, 3), np.int32)
        num = np.empty(nv, np.int32)
        means = np.empty((nv, 3), np.float32)
        covs = np.empty((nv, 3, 3), np.float32)
        self._call("get_voxel_coords", _p(coords))
        self._call("get_voxel_num_points", _p(num))
        self._call("get_voxel_means", _p(means))
        self._call("get_voxel_covs", _p(covs))
        return coords, num, means, covs

    def get_voxel_correspondences(self):
        n = self.get_num_correspondences()
        out = np.empty((n, 2), np.int32)
        self._call("get_voxel_correspondences", _p(out))
        return out

    def debug_set_voxel_hint(self, n):
        self._call("debug_set_voxel_hint", int(n))

    # ---- FastGICP (nearest target point) on the same handle ----
    def gicp_set_max_correspondence_distance(self, d):
        self._call("gicp_set_max_correspondence_distance", C.c_double(d))

    def gicp_swap_source_and_target(self):
        self._call("gicp_swap_source_and_target")

    def gicp_update_correspondences(self, T):
        t = _colmajor16(T)
        self._call("gicp_update_correspondences", _p(t))

    def gicp_compute_error(self, T, derivatives=True):
        t = _colmajor16(T)
        err = C.c_double(0)
        if derivatives:
            H = np.empty((6, 6), np.float64)
            b = np.empty(6, np.float64)
            self._call("gicp_compute_error", _p(t), _p(H), _p(b), C.byref(err))
            return err.value, H.T.copy(), b
        self._call("gicp_compute_error", _p(t), None, None, C.byref(err))
        return err.value

    def gicp_linearize(self, T):
        """FastGICP::linearize (fast_gicp_impl.hpp:159-213): update_correspondences + sums."""
        self.gicp_update_correspondences(T)
        return self.gicp_compute_error(T, True)

    def gicp_align(self, guess=None, **lm):
        g = _IDENTITY16 if guess is None else _colmajor16(guess)
        p = _lm_params(**lm) if lm else _DEFAULT_LM
        r = LmResult()
        self._call("gicp_align", _p(g), C.byref(p), C.byref(r))
        return _result_dict(r)

    def gicp_get_correspondences(self):
        out = np.empty(self.num_points("source"), np.int32)
        self._call("gicp_get_correspondences", _p(out))
        return out

    def debug_persist_aborts(self):
        n = C.c_int(0)
        self._call("debug_get_persist_aborts", C.byref(n))
        return n.value

    def debug_persist_grid(self):
        b, c = C.c_int(0), C.c_int(0)
        self._call("debug_get_persist_grid", C.byref(b), C.byref(c))
        return b.value, c.value

    def debug_table_capacity(self):
        n = C.c_int(0)
        self._call("debug_get_table_capacity", C.byref(n))
        return n.value

    def debug_skipped_points(self):
        n = C.c_int(0)
        self._call("debug_get_skipped_points", C.byref(n))
        return n.value


class VoxelGrid:
    """pcl::VoxelGrid / pcl::ApproximateVoxelGrid on the device (fvh_voxelgrid_*): same output points in the same
    order as the PCL filters the reference's callers run before registration (src/align.cpp:136-147)."""

    EXACT, APPROXIMATE = 0, 1

    def __init__(self, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.fvh_voxelgrid_create(int(device), C.byref(self._h))
        if rc != 0:
            raise FvhError("fvh_voxelgrid_create failed: %d" % rc)

    def close(self):
        if self._h:
            self._lib.fvh_voxelgrid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise FvhError("%s: %s (code %d)" % (what, (self._lib.fvh_voxelgrid_last_error(self._h) or b"").decode(), rc))

    def filter(self, xyz, leaf, method=APPROXIMATE):
        """Host array in, host array out (float32 Nx3)."""
        a = _f32(np.asarray(xyz).reshape(-1, 3))
        n = C.c_int(0)
        self._check(self._lib.fvh_voxelgrid_filter(self._h, int(method), _p(a), len(a), C.c_float(leaf), C.byref(n)), "fvh_voxelgrid_filter")
        out = np.empty((n.value, 3), np.float32)
        self._check(self._lib.fvh_voxelgrid_get_points(self._h, _p(out)), "fvh_voxelgrid_get_points")
        return out

    def filter_strided(self, buf, stride, leaf, method=APPROXIMATE):
        """Host array of n x `stride` floats (e.g. a KITTI xyzi buffer, stride 4); only the first three of each row are used."""
        a = _f32(np.asarray(buf).reshape(-1, stride))
        n = C.c_int(0)
        self._check(self._lib.fvh_voxelgrid_filter_strided(self._h, int(method), _p(a), len(a), int(stride), C.c_float(leaf), C.byref(n)), "fvh_voxelgrid_filter_strided")
        out = np.empty((n.value, 3), np.float32)
        self._check(self._lib.fvh_voxelgrid_get_points(self._h, _p(out)), "fvh_voxelgrid_get_points")
        return out

    def share_stream(self, core):
        """Run this filter on the stream of a registration handle (NDTCore / VGICPCore; None: back to its own): its output is then
        ordered before whatever that handle queues next, and filter_device(..., asynchronous=True) may return while the last kernel
        of the filter is still running. The registration handle must outlive the sharing."""
        if core is None:
            self._check(self._lib.fvh_voxelgrid_share_stream_with_ndt(self._h, None), "fvh_voxelgrid_share_stream_with_ndt")
        elif isinstance(core, NDTCore):
            self._check(self._lib.fvh_voxelgrid_share_stream_with_ndt(self._h, core.h), "fvh_voxelgrid_share_stream_with_ndt")
        else:
            self._check(self._lib.fvh_voxelgrid_share_stream_with_vgicp(self._h, core.h), "fvh_voxelgrid_share_stream_with_vgicp")

    def share_prepare_stream(self, core):
        """Run this filter on the SECOND stream of an NDTCore / VGICPCore (the one prepare_source_device works on): the next frame is
        filtered beside the LM kernel of the current one."""
        if isinstance(core, NDTCore):
            self._check(self._lib.fvh_voxelgrid_share_prepare_stream_with_ndt(self._h, core.h), "fvh_voxelgrid_share_prepare_stream_with_ndt")
        else:
            self._check(self._lib.fvh_voxelgrid_share_prepare_stream_with_vgicp(self._h, core.h), "fvh_voxelgrid_share_prepare_stream_with_vgicp")

    def filter_device(self, d_ptr, n, leaf, method=APPROXIMATE, stride=3, asynchronous=False, want_pointer=True):
        """Device pointer in (n points, `stride` floats apart); returns (device pointer to packed xyz, count) valid until the next filter call.
        asynchronous (after share_stream): the count is final on return, the points are complete in the shared stream's order only.
        want_pointer=False: (0, count) -- for callers that hand the output over with NDTCore.*_from_voxelgrid (one ABI call less per frame)."""
        m = C.c_int(0)
        fn = self._lib.fvh_voxelgrid_filter_device_async if asynchronous else self._lib.fvh_voxelgrid_filter_device
        self._check(fn(self._h, int(method), C.c_void_p(d_ptr), int(n), int(stride), C.c_float(leaf), C.byref(m)), "fvh_voxelgrid_filter_device")
        if not want_pointer:
            return 0, m.value
        ptr = C.c_void_p()
        self._check(self._lib.fvh_voxelgrid_device_points(self._h, C.byref(ptr), C.byref(m)), "fvh_voxelgrid_device_points")
        return ptr.value or 0, m.value

    def get_points(self, n):
        """Host copy of the last filter_device() result (n = the count it returned); ordered after the filter on its stream."""
        out = np.empty((int(n), 3), np.float32)
        self._check(self._lib.fvh_voxelgrid_get_points(self._h, _p(out)), "fvh_voxelgrid_get_points")
        return out

    def profile_enable(self, on=True):
        self._check(self._lib.fvh_voxelgrid_profile_enable(self._h, int(on)), "profile_enable")

    def profile_reset(self):
        self._check(self._lib.fvh_voxelgrid_profile_reset(self._h), "profile_reset")

    def profile_get(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._check(self._lib.fvh_voxelgrid_profile_get(self._h, C.byref(ms), C.byref(n)), "profile_get")
        return ms.value, n.value


class NDTCore(_Core):
    """fast_gicp::cuda::NDTCudaCore on the HIP engine."""
    _prefix = "fvh_ndt_"

    def set_distance_mode(self, mode):
        self._call("set_distance_mode", int(mode))

    def create_voxelmaps(self):
        self._call("create_voxelmaps")

    def create_target_voxelmap(self):
        self._call("create_target_voxelmap")

    def create_source_voxelmap(self):
        self._call("create_source_voxelmap")

    # ---- pipelined frame streams (include/fast_vgicp_hip.h: fvh_ndt_align_async ...) ----
    def align_async(self, guess=None, **lm):
        """Launch the LM kernel and return; align_wait() collects the result. In between only prepare_source_device() (and a VoxelGrid
        that shares the prepare stream) may be used on this handle."""
        self._g = _IDENTITY16 if guess is None else _colmajor16(guess)
        self._lm = _lm_params(**lm) if lm else _DEFAULT_LM
        self._call("align_async", _p(self._g), C.byref(self._lm))

    def align_wait(self):
        r = LmResult()
        self._call("align_wait", C.byref(r))
        return _result_dict(r)

    def prepare_source_device(self, ptr, n, stride=3):
        """The NEXT source cloud (device pointer) into the handle's prepared slot, on its second stream: widened and, for D2D, its
        voxel map built, beside whatever runs on the main stream."""
        self._call("prepare_source_device", C.c_void_p(ptr), int(n), int(stride))

    def prepare_source(self, xyz):
        """The same for a host cloud (N x 3 float32): consumed before the call returns."""
        a = np.ascontiguousarray(xyz, np.float32)
        self._call("prepare_source", _p(a), len(a), 3)

    # ---- the filter's last ApproximateVoxelGrid output becomes the cloud without a copy (fvh_ndt_*_from_voxelgrid) ----
    def set_source_cloud_from_voxelgrid(self, vg):
        self._call("set_source_cloud_from_voxelgrid", vg._h)

    def set_target_cloud_from_voxelgrid(self, vg):
        self._call("set_target_cloud_from_voxelgrid", vg._h)

    def prepare_source_from_voxelgrid(self, vg):
        self._call("prepare_source_from_voxelgrid", vg._h)

    def adopt_prepared_source(self):
        self._call("adopt_prepared_source")

    def debug_set_voxel_hint(self, which, n):
        self._call("debug_set_voxel_hint", 0 if which == "source" else 1, int(n))

    def set_source_tile(self, rank, nranks):
        """Evaluate tile `rank` of `nranks` of the source only (P2D: points in Morton order, D2D: source voxels ranked by key): linearize /
        compute_error return PARTIAL sums, to be added over the ranks; nranks = 1 switches it off."""
        self._call("set_source_tile", int(rank), int(nranks))

    def get_num_voxels(self, which):
        n = C.c_int(0)
        self._call("get_num_voxels", 0 if which == "source" else 1, C.byref(n))
        return n.value

    def get_voxel_correspondences(self):
        """(n, 2): (source element, target voxel index of get_voxelmap("target")); the source element is a point index (P2D) or an
        index into get_voxelmap("source") (D2D)."""
        n = self.get_num_correspondences()
        out = np.empty((max(n, 1), 2), np.int32)
        self._call("get_voxel_correspondences", _p(out))
        return out[:n]

    def get_voxelmap(self, which):
        w = 1 if which == "target" else 0
        n = C.c_int(0)
        self._call("get_num_voxels", w, C.byref(n))
        nv = n.value
        coords = np.empty((nv, 3), np.int32)
        num = np.empty(nv, np.int32)
        means = np.empty((nv, 3), np.float32)
        covs = np.empty((nv, 3, 3), np.float32)
        self._call("get_voxels", w, _p(coords), _p(num), _p(means), _p(covs))
        return coords, num, means, covs
