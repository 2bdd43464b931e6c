"""Per-iteration times of the pipelined VGICP loop on the synthetic 100k pair (debug)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fast_gicp_amd import capi, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rbf = (sys.argv[2] == "rbf") if len(sys.argv) > 2 else True
stages = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tgt, src, _ = workloads.synthetic_pair(n, n, seed=42)
gpu = torch.device("cuda", 0)
d = [torch.from_numpy(tgt).to(gpu).contiguous(), torch.from_numpy(src).to(gpu).contiguous()]
c = capi.VGICPCore(0)
c.set_resolution(0.5); c.set_neighbor_search_method(capi.DIRECT27); c.set_kernel_params(0.5, 2.5)
def cov(which):
    if rbf: getattr(c, "calculate_%s_covariances_rbf" % which)(capi.REG_PLANE)
    else:
        getattr(c, "find_%s_neighbors" % which)(20); getattr(c, "calculate_%s_covariances" % which)(capi.REG_PLANE)
c.set_target_cloud_device(d[0].data_ptr(), n, 3); cov("target"); c.create_target_voxelmap()
c.set_source_cloud_device(d[1].data_ptr(), n, 3); cov("source")
for _ in range(3):
    c.align(); c.swap_source_and_target()
nxt = 1
ts = []
for it in range(int(os.environ.get('ITERS', '14'))):
    t0 = time.perf_counter()
    c.align_async()
    t1 = time.perf_counter()
    c.prepare_source_device(d[nxt].data_ptr(), n, 3, 20, capi.REG_PLANE, rbf, stages)
    t2 = time.perf_counter()
    r = c.align_wait()
    t3 = time.perf_counter()
    c.swap_source_and_target(); c.adopt_prepared_source()
    if stages < 2: (c.calculate_source_covariances_rbf if rbf else c.calculate_source_covariances)(capi.REG_PLANE)
    t4 = time.perf_counter()
    nxt = 1 - nxt
    ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, r["num_launches"], c.debug_persist_aborts()))
for i, t in enumerate(ts):
    if sum(t[:4]) > 2e-3:
        print("slow iteration %d: async %.0f us  prepare %.0f us  wait %.0f us  swap+adopt %.0f us  launches %d aborts %d" % (i, t[0] * 1e6, t[1] * 1e6, t[2] * 1e6, t[3] * 1e6, t[4], t[5]))
print("mean iteration %.1f us, aborts %d" % (np.mean([sum(t[:4]) for t in ts]) * 1e6, ts[-1][5]))
for t in ts[:int(os.environ.get('SHOW', '14'))]:
    print("async %.0f us  prepare %.0f us  wait %.0f us  swap+adopt %.0f us  launches %d aborts %d" % (t[0] * 1e6, t[1] * 1e6, t[2] * 1e6, t[3] * 1e6, t[4], t[5]))
