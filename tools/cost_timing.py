"""Debug: epilogue timeline of cost_kernel (needs a library built with FVH_EXTRA_HIPCC_FLAGS=-DFVH_COST_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_gicp_amd import capi, preprocess  # noqa: E402

# the instrumented build lives next to the product library: FVH_EXTRA_HIPCC_FLAGS=-DFVH_COST_TIMING python fast_gicp_amd/build.py,
# then move the result to fast_gicp_amd/lib/dbg/ and rebuild the product library without the flag
_dbg = os.path.join(os.path.dirname(capi.lib_path()), "dbg", "libfast_vgicp_hip.so")
if os.path.exists(_dbg):
    capi._LIB = C.CDLL(_dbg)
    capi._LIB.fvh_vgicp_last_error.restype = C.c_char_p
    capi._LIB.fvh_ndt_last_error.restype = C.c_char_p
L = capi.load()
tgt, src = preprocess.bundled_pair(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data"))
buf = (C.c_ulonglong * 16)()


def stamps(label, fn, reps=30):
    rows = []
    for _ in range(reps):
        L.fvh_debug_cost_timing(None, 1)
        fn()
        L.fvh_debug_cost_timing(buf, 0)
        v = np.array(list(buf), np.float64)
        rows.append([(v[i] - v[9]) * 10.0 for i in range(1, 8)] + [v[8]])  # ns, relative to the last workgroup's own start
    r = np.median(np.array(rows), axis=0)
    print("%-28s grid %4d | main %6.0f  blockred %6.0f  ticket1 %6.0f  groupred+ticket2 %6.0f  final-sum+ldcopy %6.0f  lmstep %6.0f  writeback %6.0f | total %6.0f ns" % (
        label, r[7], r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], r[6]))


for search, name in ((capi.DIRECT27, "vgicp D27"), (capi.DIRECT7, "vgicp D7"), (capi.DIRECT1, "vgicp D1")):
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
    # max_iterations=1 -> the align is 2 launches (linearize, fused trial); the stamps are those of the last one
    stamps(name + " align(1 iter)", lambda: c.align(max_iterations=1))
    stamps(name + " linearize (host)", lambda: c.linearize(np.eye(4)))
    c.close()
n = capi.NDTCore(0)
n.set_target_cloud(tgt); n.set_source_cloud(src)
stamps("ndt d2d align(1 iter)", lambda: n.align(max_iterations=1))
