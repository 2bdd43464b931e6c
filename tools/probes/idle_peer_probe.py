import os, sys, time, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "idle":
    from fast_gicp_amd import capi
    c = capi.VGICPCore(0)
    print("idle ready", flush=True)
    time.sleep(float(sys.argv[2]))
    sys.exit(0)
from fast_gicp_amd import capi, workloads
tgt, src, _ = workloads.synthetic_pair(100000, 100000, seed=44, extent=150.0)
def run(label):
    c = capi.VGICPCore(0)
    c.set_resolution(0.5); c.set_neighbor_search_method(capi.DIRECT7)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
    for _ in range(3): c.align()
    t0 = time.perf_counter()
    for _ in range(20): r = c.align()
    print(label, "ms/align %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), "aborts", c.debug_persist_aborts(), "launches", r["num_launches"], flush=True)
    c.close()
run("alone")
p = subprocess.Popen([sys.executable, __file__, "idle", "20"], stdout=subprocess.PIPE, text=True)
p.stdout.readline()
run("with idle process")
os.environ["X"] = "1"
p.wait()
run("alone again")
