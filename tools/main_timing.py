"""Debug: timeline of wave 0 of every workgroup through the main loop of the persistent LM kernel, per trip (library built
with -DFVH_COST_TIMING: tools/build_variants.py timing="-DFVH_COST_TIMING", run with FVH_LIB_PATH=.../variants/timing/...).
The stamps wait for everything in flight first, so they serialise what the scheduler overlaps: upper bounds per segment."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess  # noqa: E402

L = capi.load()
if "--ndt" in sys.argv:  # the LiDAR-stream configuration: NDT D2D, DIRECT7, two consecutive simulated frames after ApproximateVoxelGrid 0.25
    from fast_gicp_amd import workloads
    vg = capi.VoxelGrid(0)
    c = capi.NDTCore(0)
    c.set_distance_mode(capi.NDT_D2D); c.set_neighbor_search_method(capi.DIRECT7); c.set_resolution(1.0)
    c.set_target_cloud(vg.filter(workloads.lidar_frame(3), 0.25, vg.APPROXIMATE))
    c.set_source_cloud(vg.filter(workloads.lidar_frame(4), 0.25, vg.APPROXIMATE))
elif "--synth1m" in sys.argv:  # BASELINE configs[4] on one GPU: 1M-point map <-> 100k-point scan, res 0.5, DIRECT7 (768 workgroups, two items per thread)
    from fast_gicp_amd import workloads
    tgt, src, _ = workloads.synthetic_pair(1_000_000, 100_000, seed=44, extent=150.0)
    c = capi.VGICPCore(0)
    c.set_resolution(0.5); c.set_neighbor_search_method(capi.DIRECT7)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
else:
    tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(capi.DIRECT27)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
for rep in range(5):
    c.align()
L.fvh_debug_main_timing(None, 1)
L.fvh_debug_persist_timing(None, 1)
r = c.align()
m = np.zeros((16, 512, 12), np.uint64)  # (the first 512 workgroups only)
pt = np.zeros((16, 512, 12), np.uint64)
L.fvh_debug_main_timing(m.ctypes.data_as(C.c_void_p), 0)
L.fvh_debug_persist_timing(pt.ctypes.data_as(C.c_void_p), 0)
trips = r["num_error_evals"] + 1
live = m[0, :, 0] > 0
print("workgroups:", int(live.sum()), "trips:", trips)
names = ["trip start -> loop entry", "source element, stored ids, offsets arrived", "poses (LDS), R C R^T, q, voxel coord", "first probes + old records arrived",
         "trial error of the old ids", "ids resolved (+ stores retired)", "records of changed ids arrived", "hit terms of the new ids", "item sums + butterfly"]
for t in range(min(trips, 16)):
    x = m[t, live].astype(np.float64) / 100.0
    x = x[:, :9]
    s0 = pt[t, live, 0].astype(np.float64) / 100.0
    seg = np.concatenate([(x[:, 0] - s0)[:, None], np.diff(x, axis=1)], axis=1)
    tot = x[:, 8] - s0
    print("trip %d: total main loop med %.2f  p90 %.2f  max %.2f us" % (t, np.median(tot), np.quantile(tot, 0.9), tot.max()))
    for k in range(9):
        print("    %-46s med %5.2f  p90 %5.2f  max %5.2f" % (names[k], np.median(seg[:, k]), np.quantile(seg[:, k], 0.9), seg[:, k].max()))
    if t == 2:  # who is slow? by XCD (workgroup % 8) and by workgroup index
        b = np.nonzero(live)[0]
        for xcd in range(8):
            print("      xcd %d: med %.2f max %.2f" % (xcd, np.median(tot[b % 8 == xcd]), tot[b % 8 == xcd].max()))
        order = np.argsort(-tot)[:12]
        print("      slowest workgroups:", [(int(b[i]), round(float(tot[i]), 2)) for i in order])
        q = len(b) // 4
        for qi in range(4):
            sl = slice(qi * q, (qi + 1) * q)
            print("      workgroups %3d..%3d: med %.2f" % (b[sl][0], b[sl][-1], np.median(tot[sl])))
