"""Debug probe: the sequence of tests/test_gpu_robustness.py::test_concurrent_handles_share_the_coresident_slots with diagnostics."""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess
tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
def make():
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(0)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    return c
cores = [make() for _ in range(4)]
ref = cores[0].align()
print("ref launches", ref["num_launches"], "grid", cores[0].debug_persist_grid(), "pool", capi.debug_slot_pool(0))
for i, c in enumerate(cores):
    r = c.align()
    print("core", i, "seq align equal to ref:", np.array_equal(r["T"], ref["T"]), np.abs(r["T"] - ref["T"]).max(), "grid", c.debug_persist_grid(), r["num_launches"])
def loop(c):
    for _ in range(30):
        c.align()
th = [threading.Thread(target=loop, args=(c,)) for c in cores]
[t.start() for t in th]; [t.join() for t in th]
print("pool after threads", capi.debug_slot_pool(0))
n = 0
while capi.debug_slot_pool(0)[2] > 1 and n < 300:
    r = cores[0].align(); n += 1
print("decay aligns", n, "pool", capi.debug_slot_pool(0))
for k in range(3):
    r = cores[0].align()
    print("veteran equal to ref:", np.array_equal(r["T"], ref["T"]), np.abs(r["T"] - ref["T"]).max(), "grid", cores[0].debug_persist_grid(), r["num_launches"], r["num_linearize"], r["num_error_evals"])
lone = make()
for k in range(3):
    r = lone.align()
    print("lone equal to ref:", np.array_equal(r["T"], ref["T"]), np.abs(r["T"] - ref["T"]).max(), "grid", lone.debug_persist_grid(), r["num_launches"])
covs = [c.get_covariances("source") for c in cores] + [lone.get_covariances("source")]
print("source covariances identical across handles:", [np.array_equal(covs[0], x) for x in covs[1:]])
covt = [c.get_covariances("target") for c in cores] + [lone.get_covariances("target")]
print("target covariances identical across handles:", [np.array_equal(covt[0], x) for x in covt[1:]])
