"""Exact 1-NN search (getFitnessScore / FastGICP correspondences: kernels_cov.hpp nn1_group_kernel) timed with HIP events through the C ABI:
17k x 17k (bundled pair), 100k x 100k, 100k queries in the 1M-point map; and the FastGICP registration loop on the bundled pair.
(four queries per wave, one per 16-lane row; boxes nearest first)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess, workloads  # noqa: E402


def timed(name, tgt, src, T, reps=20):
    c = capi.VGICPCore(0)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    for _ in range(3):
        f = c.fitness_score(T)
    c.profile_enable(True); c.profile_reset()
    for _ in range(reps):
        f = c.fitness_score(T)
    ms, n = c.profile_get("fitness")
    print("%-28s %7d queries in %8d points: search + reduce %8.2f us (fitness %.6f)" % (name, len(src), len(tgt), ms / n * 1e3, f))
    c.close()


tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
T = np.loadtxt(os.path.join(ROOT, "data", "relative.txt"))
timed("bundled 17k", tgt, src, T)
timed("bundled 17k (identity)", tgt, src, np.eye(4))
t1, s1, T1 = workloads.synthetic_pair(100000, 100000)
timed("synth 100k x 100k", t1, s1, T1)
if "--no-1m" not in sys.argv:
    from bench import make_workload
    t2, s2, _, _ = make_workload("synth1m")
    timed("synth 100k in 1M", t2, s2, np.eye(4), reps=10)

# FastGICP loop (align.cpp 100times_reuse shape: swap, new source, align)
c = capi.VGICPCore(0)
clouds = [tgt, src]
c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances()
c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances()
r = c.gicp_align()
nxt = 0
for it in range(110):
    if it == 10:
        c.synchronize(); t0 = time.perf_counter()
    c.gicp_swap_source_and_target()
    c.set_source_cloud(clouds[nxt]); c.find_source_neighbors(20); c.calculate_source_covariances()
    r = c.gicp_align()
    nxt = 1 - nxt
c.synchronize()
el = time.perf_counter() - t0
print("FastGICP 100times_reuse (C ABI, host clouds in): %.2f ms per 100 registrations, %d launches in the last align, converged %s" % (el * 1e3, r["num_launches"], r["converged"]))
c.profile_enable(True); c.profile_reset()
c.gicp_swap_source_and_target(); c.set_source_cloud(clouds[nxt]); c.find_source_neighbors(20); c.calculate_source_covariances(); r = c.gicp_align()
print("  per class (ms total, launches):", {k: c.profile_get(k) for k in ("gicp_nn", "cost", "knn", "sort", "cov")})
