"""GICP (nearest-point) registration time through the C ABI with the host LM loop (FastGICP flow)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess, distributed as D  # noqa: E402

tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
c = capi.VGICPCore(0)
c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances()
c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
lsq = D.ShardedLsq(lambda T: c.gicp_linearize(T), lambda T: c.gicp_compute_error(T, derivatives=False), lambda v: v)
lsq.align()
n = 50
t = time.perf_counter()
for _ in range(n):
    r = lsq.align()
el = (time.perf_counter() - t) / n
c.profile_enable(True); c.profile_reset(); lsq.align()
print("gicp host-LM align %.3f ms, t = %s, nn %s, cost %s" % (el * 1e3, np.round(r["T"][:3, 3], 6), c.profile_get("gicp_nn"), c.profile_get("cost")))
