"""Debug: per-trip timeline of the persistent LM kernel (library built with FVH_EXTRA_HIPCC_FLAGS=-DFVH_COST_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess  # noqa: E402

L = capi.load()
tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
buf = np.zeros((16, 512, 12), np.uint64)
c = capi.VGICPCore(0)
c.set_neighbor_search_method(capi.DIRECT27)
c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
for rep in range(5):
    c.align()
L.fvh_debug_persist_timing(None, 1)
r = c.align()
L.fvh_debug_persist_timing(buf.ctypes.data_as(C.c_void_p), 0)
trips = r["num_error_evals"] + 1
v = buf.astype(np.float64)
live = v[0, :, 0] > 0
nb = int(live.sum())
t0 = v[0, live, 0].min()
print("workgroups:", nb, " trips:", trips)
print("opener path per trip [us after the last arrival]: group-rows reduced (its own group-last) | top atomic returned | final reduce + state in LDS | LM step + broadcast issued")
for t in range(min(trips, 16)):
    x = (v[t, live] - t0) / 100.0
    o = np.argmax(v[t, live, 6])  # the opener
    la = x[:, 2].max()
    print("  trip %d: arrive(own) %.2f  last arrival %.2f | %.2f | %.2f | %.2f | %.2f   [sums->LDS %.2f, lm_step %.2f, fill+sync %.2f, bcast stores %.2f]" % (
        t, x[o, 2], la, x[o, 5] - la, x[o, 6] - la, x[o, 7] - la, x[o, 3] - la, x[o, 8] - x[o, 7], x[o, 9] - x[o, 8], x[o, 10] - x[o, 9], x[o, 3] - x[o, 10]))
print("trip | start(min/max)   main_end(med/max)  arrive(max)   open    seen(min/med/max)      lm_done(med/max)   [us since first workgroup start]")
for t in range(min(trips, 16)):
    x = (v[t, live] - t0) / 100.0
    op = x[:, 3][v[t, live, 3] > 0]
    print("%4d | %7.2f %7.2f   %8.2f %8.2f   %9.2f   %7.2f   %7.2f %7.2f %7.2f   %8.2f %8.2f" % (
        t, x[:, 0].min(), x[:, 0].max(), np.median(x[:, 1]), x[:, 1].max(), x[:, 2].max(), op.max() if len(op) else -1, x[:, 4].min(), np.median(x[:, 4]), x[:, 4].max(),
        np.median(x[:, 5]), x[:, 5].max()))
