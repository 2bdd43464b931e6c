"""Debug: per-query timeline of knn_tiled1_kernel (library built with FVH_EXTRA_HIPCC_FLAGS=-DFVH_KNN_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess  # noqa: E402

L = capi.load()
tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
if len(sys.argv) > 1 and sys.argv[1] == "target":  # the other cloud of the pair (its search takes 60 us against 33)
    src = tgt
c = capi.VGICPCore(0)
c.set_source_cloud(src)
for _ in range(3):
    c.set_source_cloud(src); c.find_source_neighbors(20); c.synchronize()
buf = np.zeros((32768, 8), np.uint64)
L.fvh_debug_knn_timing(buf.ctypes.data_as(C.c_void_p))
v = buf[:len(src)].astype(np.float64)
t = v[:, :5]
d = np.diff(t, axis=1) / 100.0  # us
print("queries %d  kernel span %.1f us" % (len(src), (t[:, 4].max() - t[:, 0].min()) / 100.0))
print("per query [us] median / p90:  seed loads %.2f / %.2f   bitonic sort %.2f / %.2f   2 neighbour tiles %.2f / %.2f   culled sweep %.2f / %.2f   total %.2f / %.2f" % (
    np.median(d[:, 0]), np.percentile(d[:, 0], 90), np.median(d[:, 1]), np.percentile(d[:, 1], 90), np.median(d[:, 2]), np.percentile(d[:, 2], 90),
    np.median(d[:, 3]), np.percentile(d[:, 3], 90), np.median(t[:, 4] - t[:, 0]) / 100, np.percentile(t[:, 4] - t[:, 0], 90) / 100))
print("tiles merged per query: mean %.1f  insertions: mean %.1f p90 %.0f" % (v[:, 5].mean(), v[:, 6].mean(), np.percentile(v[:, 6], 90)))
start = np.sort(t[:, 0] - t[:, 0].min()) / 100.0
print("wave starts [us]: 10%% %.1f  50%% %.1f  90%% %.1f  last %.1f" % (start[len(start) // 10], start[len(start) // 2], start[9 * len(start) // 10], start[-1]))
dur = (t[:, 4] - t[:, 0]) / 100.0
end = (t[:, 4] - t[:, 0].min()) / 100.0
o = np.argsort(-end)[:12]
print("last finishers: query(sorted pos) start end dur tiles insertions")
for i in o:
    print("  %6d  %6.1f %6.1f %6.1f  %4d %5d" % (i, (t[i, 0] - t[:, 0].min()) / 100.0, end[i], dur[i], v[i, 5], v[i, 6]))
print("duration percentiles [us]: 50 %.1f 90 %.1f 99 %.1f 99.9 %.1f max %.1f" % tuple(np.percentile(dur, [50, 90, 99, 99.9, 100])))
print("tiles percentiles: 50 %d 90 %d 99 %d max %d;  insertions: 50 %d 90 %d 99 %d max %d" % (tuple(np.percentile(v[:, 5], [50, 90, 99, 100])) + tuple(np.percentile(v[:, 6], [50, 90, 99, 100]))))
