"""Can the pre-LM chain of the NEXT registration (pack, sort, k-NN, covariances, voxel-map build) run beside the persistent LM kernel of the current one?
Two handles, two host threads: A aligns in a loop, B prepares clouds in a loop. Rates alone and together."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fast_gicp_amd import capi, preprocess

tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
gpu = torch.device("cuda", 0)
d = [torch.from_numpy(tgt).to(gpu).contiguous(), torch.from_numpy(src).to(gpu).contiguous()]
a = capi.VGICPCore(0)
a.set_neighbor_search_method(capi.DIRECT27)
a.set_target_cloud_device(d[0].data_ptr(), len(tgt), 3); a.find_target_neighbors(20); a.calculate_target_covariances(); a.create_target_voxelmap()
a.set_source_cloud_device(d[1].data_ptr(), len(src), 3); a.find_source_neighbors(20); a.calculate_source_covariances()
b = capi.VGICPCore(0)
b.set_neighbor_search_method(capi.DIRECT27)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b.set_engine_params(sort_mode=mode)
stop = [False]

def run_a(n):
    for _ in range(5): a.align()
    t = time.perf_counter()
    for _ in range(n): a.align()
    return (time.perf_counter() - t) / n * 1e6

def prep_b():
    b.set_target_cloud_device(d[1].data_ptr(), len(src), 3); b.find_target_neighbors(20); b.calculate_target_covariances(); b.create_target_voxelmap(); b.synchronize()

def run_b(n):
    for _ in range(5): prep_b()
    t = time.perf_counter()
    for _ in range(n): prep_b()
    return (time.perf_counter() - t) / n * 1e6

print("alone: align %.1f us, prepare chain (sort mode %d) %.1f us" % (run_a(300), mode, run_b(300)))
res = {}
def loop_b():
    k = 0; t = time.perf_counter()
    while not stop[0]:
        prep_b(); k += 1
    res["b"] = (time.perf_counter() - t) / max(k, 1) * 1e6
th = threading.Thread(target=loop_b); th.start()
time.sleep(0.05)
ta = run_a(1000)
stop[0] = True; th.join()
print("together: align %.1f us, prepare chain %.1f us; aborts %d" % (ta, res["b"], a.debug_persist_aborts()))
