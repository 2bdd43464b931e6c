"""Debug: per-trip timeline of the persistent LM kernel (library built with FVH_EXTRA_HIPCC_FLAGS=-DFVH_COST_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess  # noqa: E402

L = capi.load()
buf = np.zeros((16, 512, 12), np.uint64)
if "--ndt" in sys.argv:  # the LiDAR-stream configuration: NDT D2D, DIRECT7, two consecutive simulated frames after ApproximateVoxelGrid 0.25
    from fast_gicp_amd import workloads
    vg = capi.VoxelGrid(0)
    c = capi.NDTCore(0)
    c.set_distance_mode(capi.NDT_D2D); c.set_neighbor_search_method(capi.DIRECT7); c.set_resolution(1.0)
    f0, f1 = workloads.lidar_frame(3), workloads.lidar_frame(4)
    c.set_target_cloud(vg.filter(f0, 0.25, vg.APPROXIMATE))
    c.set_source_cloud(vg.filter(f1, 0.25, vg.APPROXIMATE))
else:
    tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(capi.DIRECT27)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
for rep in range(5):
    c.align()
L.fvh_debug_persist_timing(None, 1)
lm = np.zeros((16, 8), np.uint64)
has_lm = hasattr(L, "fvh_debug_lm_timing")
if has_lm:
    L.fvh_debug_lm_timing(None, 1)
r = c.align()
L.fvh_debug_persist_timing(buf.ctypes.data_as(C.c_void_p), 0)
if has_lm:
    L.fvh_debug_lm_timing(lm.ctypes.data_as(C.c_void_p), 0)
    print("LM step, shader cycles per stage (the stamps wait for everything in flight: upper bounds): state + decision | H, b to registers | (-) | LDL^T + solves | se3_exp | convergence test | tail")
    for k in range(16):
        if lm[k, 0] == 0:
            break
        d = np.diff(lm[k, :7].astype(np.int64))
        print("  call %2d: %s  total %d" % (k, " ".join("%5d" % v for v in d), int(lm[k, 6] - lm[k, 0])))
trips = r["num_error_evals"] + 1
v = buf.astype(np.float64)
live = v[0, :, 0] > 0
nb = int(live.sum())
t0 = v[0, live, 0].min()
print("workgroups:", nb, " trips:", trips)
# stamps per workgroup and trip: 0 trip start, 1 main loop end, 2 row published, 5 group row published (collectors), 6 opener holds the
# group rows, 7 final sums, 8 LM step starts, 9 LM step done, 10 payload built, 3 broadcast stores issued (opener), 4 broadcast seen
print("trip | start min/max | main end med/max | last row pub | group rows (max) +d | opener has sums +d | LM step | bcast issued +d | seen med/max +d | next start max")
tot = []
for t in range(min(trips, 16)):
    x = (v[t, live] - t0) / 100.0
    raw = v[t, live]
    o = int(np.argmax(raw[:, 9]))  # the opener
    last_pub = x[:, 2].max()
    coll = x[:, 5][raw[:, 5] > 0]
    g = coll.max() if len(coll) else float("nan")
    seen = x[:, 4][raw[:, 4] > 0]
    print("%4d | %6.2f %6.2f | %6.2f %6.2f | %7.2f | %7.2f %+5.2f | %7.2f %+5.2f | %5.2f | %7.2f %+5.2f | %7.2f %7.2f %+5.2f" % (
        t, x[:, 0].min(), x[:, 0].max(), np.median(x[:, 1]), x[:, 1].max(), last_pub, g, g - last_pub, x[o, 7], x[o, 7] - last_pub, x[o, 9] - x[o, 8], x[o, 3], x[o, 3] - x[o, 9],
        np.median(seen) if len(seen) else float("nan"), seen.max() if len(seen) else float("nan"), (seen.max() - x[o, 3]) if len(seen) else float("nan")))
    if raw[o, 11] > 0:
        print("       LM step: %d shader cycles in %.2f us -> %.2f GHz" % (int(raw[o, 11]), x[o, 9] - x[o, 8], raw[o, 11] / (x[o, 9] - x[o, 8]) / 1e3))
    tot.append((x[:, 1].max() - x[:, 0].min(), (seen.max() if len(seen) else x[o, 3]) - x[:, 1].max()))
tot = np.array(tot)
print("per trip (mean over trips): main loop incl. skew %.2f us, epilogue (last main end -> broadcast seen by all) %.2f us" % (tot[:, 0].mean(), tot[:-1, 1].mean() if len(tot) > 1 else tot[:, 1].mean()))

# who finishes the main loop late? (workgroups 0..255 are the first one on their CU under the observed dispatch order, the rest the second)
idx = np.nonzero(live)[0]
for t in range(1, min(trips, 4)):
    x = (v[t, live] - t0) / 100.0
    dur = x[:, 1] - x[:, 0]
    for name, sel in (("blockIdx < 256", idx < 256), ("blockIdx >= 256", idx >= 256)):
        if sel.any():
            q = np.percentile(dur[sel], [10, 50, 90, 100])
            e = np.percentile(x[sel, 1] - x[:, 0].min(), [50, 90, 100])
            print("trip %d main loop of %-16s (%3d workgroups): duration p10 %.2f p50 %.2f p90 %.2f max %.2f us; end after the trip's first start: p50 %.2f p90 %.2f max %.2f" % (
                t, name, int(sel.sum()), q[0], q[1], q[2], q[3], e[0], e[1], e[2]))
