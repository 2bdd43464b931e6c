import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from fast_gicp_amd import capi, workloads
tgt, src, _ = workloads.synthetic_pair(1_000_000, 100_000, seed=44, extent=150.0)
c = capi.VGICPCore(0)
c.set_resolution(0.5); c.set_neighbor_search_method(capi.DIRECT7)
c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
for n in (100000, 98304, 98000, 90000, 100000):
    s = np.ascontiguousarray(src[:n])
    c.set_source_cloud(s); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
    for _ in range(3): r = c.align()
    c.profile_reset(); c.profile_enable(2)
    for _ in range(20): r = c.align()
    c.profile_enable(0)
    ms, k = c.profile_get("cost")
    print("scan points %6d  items %6d  LM launch %.1f us  evals %d grid %s" % (n, n * 2, ms / k * 1e3, r["num_linearize"] + r["num_error_evals"], c.debug_persist_grid()))
