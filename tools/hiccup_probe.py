"""Are there periodic multi-millisecond stalls in a long run of registrations? Sequential reuse loop and the pipelined one, per-iteration wall times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fast_gicp_amd import capi, preprocess

if len(sys.argv) > 2 and sys.argv[2].startswith("ndt"):
    # the LiDAR frame stream (bench.py --workload lidar_stream): raw 118k-point frames, ApproximateVoxelGrid 0.25 on the device, NDT D2D DIRECT7
    from tests import util
    iters = int(sys.argv[1]); mode = sys.argv[2]
    gpu = torch.device("cuda", 0)
    raw = [util.lidar_frame(i) for i in range(8)]
    d_raw = [torch.from_numpy(f).to(gpu).contiguous() for f in raw]
    vg, c = capi.VoxelGrid(0), capi.NDTCore(0)
    c.set_distance_mode(1); c.set_neighbor_search_method(1); c.set_resolution(1.0)
    ptr, n = vg.filter_device(d_raw[0].data_ptr(), len(raw[0]), 0.25)
    c.set_target_cloud_device(ptr, n, 3)
    ts = np.zeros(iters)
    if mode == "ndt_seq":
        vg.share_stream(c)
        for it in range(iters):
            i = 1 + it % 7
            t0 = time.perf_counter()
            ptr, n = vg.filter_device(d_raw[i].data_ptr(), len(raw[i]), 0.25, asynchronous=True)
            c.set_source_cloud_from_voxelgrid(vg)
            c.align()
            c.swap_source_and_target()
            ts[it] = time.perf_counter() - t0
    else:
        vg.share_prepare_stream(c)
        ptr, n = vg.filter_device(d_raw[1].data_ptr(), len(raw[1]), 0.25, asynchronous=True)
        c.prepare_source_device(ptr, n, 3)
        for it in range(iters):
            i = 1 + (it + 1) % 7
            t0 = time.perf_counter()
            c.adopt_prepared_source()
            c.align_async()
            ptr, n = vg.filter_device(d_raw[i].data_ptr(), len(raw[i]), 0.25, asynchronous=True)
            c.prepare_source_from_voxelgrid(vg)
            c.align_wait()
            c.swap_source_and_target()
            ts[it] = time.perf_counter() - t0
    print(mode, "iterations", iters, "median %.1f us" % (np.median(ts) * 1e6), "mean %.1f us" % (ts.mean() * 1e6))
    sys.exit(0)
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
