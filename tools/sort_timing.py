"""Debug: phase timeline of the cooperative small sort (library built with -DFVH_SORT_TIMING:
    python tools/build_variants.py sort_timing="-DFVH_SORT_TIMING"; FVH_LIB_PATH=fast_gicp_amd/lib/variants/sort_timing/libfast_vgicp_hip.so python tools/sort_timing.py)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess  # noqa: E402

L = capi.load()
tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
c = capi.VGICPCore(0)
NAMES = ["start", "keys", "p0 have elems", "p0 totals pub", "p0 row seen (t0)", "p0 cursors", "p0 scatter issued", "p1 have elems", "p1 totals pub", "p1 row seen (t0)", "p1 cursors",
         "p1 scatter issued", "-", "tile boxes", "p0 all rows seen (barrier)"]
for rep in range(6):
    c.set_source_cloud(src)
    c.find_source_neighbors(20)
    c.synchronize()
buf = np.zeros((32, 16), np.uint64)
assert L.fvh_debug_sort_timing(buf.ctypes.data_as(C.c_void_p)) == 0
v = buf.astype(np.float64)
t0 = v[:, 0].min()
x = (v - t0) / 100.0
print("stamp (us after the first workgroup's start)      min    med    max")
for k, nm in enumerate(NAMES):
    if nm == "-":
        continue
    col = x[:, k]
    print("%2d %-22s %6.2f %6.2f %6.2f" % (k, nm, col.min(), np.median(col), col.max()))
print("failed polls of wave 0 up to the pass-0 barrier, per workgroup:", buf[:, 15].astype(int).tolist())
