"""Are there periodic multi-millisecond stalls in a long run of registrations? Sequential reuse loop and the pipelined one, per-iteration wall times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fast_gicp_amd import capi, preprocess

tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
gpu = torch.device("cuda", 0)
cl = [tgt, src]
d = [torch.from_numpy(c).to(gpu).contiguous() for c in cl]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
mode = sys.argv[2] if len(sys.argv) > 2 else "seq"
c = capi.VGICPCore(0)
c.set_neighbor_search_method(capi.DIRECT27)
c.set_target_cloud_device(d[0].data_ptr(), len(cl[0]), 3); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
c.set_source_cloud_device(d[1].data_ptr(), len(cl[1]), 3); c.find_source_neighbors(20); c.calculate_source_covariances()
c.align()
nxt = 0
ts = np.zeros(iters)
if mode == "seq":
    for it in range(iters):
        t0 = time.perf_counter()
        c.swap_source_and_target()
        c.set_source_cloud_device(d[nxt].data_ptr(), len(cl[nxt]), 3); c.find_source_neighbors(20); c.calculate_source_covariances()
        c.align()
        ts[it] = time.perf_counter() - t0
        nxt = 1 - nxt
elif mode == "align_only":
    for it in range(iters):
        t0 = time.perf_counter()
        c.align()
        ts[it] = time.perf_counter() - t0
else:
    for it in range(iters):
        t0 = time.perf_counter()
        c.align_async()
        c.prepare_source_device(d[nxt].data_ptr(), len(cl[nxt]), 3, 20, capi.REG_PLANE, False, 2)
        c.align_wait()
        c.swap_source_and_target(); c.adopt_prepared_source()
        ts[it] = time.perf_counter() - t0
        nxt = 1 - nxt
slow = np.nonzero(ts > 2e-3)[0]
print(mode, "iterations", iters, "median %.1f us" % (np.median(ts) * 1e6), "mean %.1f us" % (ts.mean() * 1e6), "aborts", c.debug_persist_aborts())
print("  iterations over 2 ms:", [(int(i), round(ts[i] * 1e3, 1)) for i in slow][:20])
