#!/usr/bin/env python
"""Per-phase kernel timings of the engine (HIP events on the handle's stream via the profile API)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bundled17k")
    ap.add_argument("--search", default="DIRECT27")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--precision", default="fp64")
    ap.add_argument("--skip-knn", action="store_true")
    args = ap.parse_args()
    from fast_gicp_amd import capi
    import bench
    tgt, src, res, desc = bench.make_workload(args.workload)
    c = capi.VGICPCore(0)
    c.set_resolution(res)
    c.set_neighbor_search_method(getattr(capi, args.search))
    c.set_kernel_params(0.5, 2.5)
    c.set_precision(capi.COMPUTE_FP32 if args.precision == "fp32" else capi.COMPUTE_FP64)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    print("workload:", desc, "search", args.search, "FVH_COST_TARGET_ITEMS=%s FVH_COST_MAX_BLOCKS=%s" % (os.environ.get("FVH_COST_TARGET_ITEMS"), os.environ.get("FVH_COST_MAX_BLOCKS")))

    def timed(label, cls, fn, reps=args.reps):
        fn()
        c.profile_reset(); c.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        c.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e6
        c.profile_enable(False)
        ms, n = c.profile_get(cls)
        print("%-34s kernel %9.2f us/launch (%d launches)   wall %9.2f us/call" % (label, ms / max(n, 1) * 1e3, n, wall))

    if args.skip_knn:
        c.calculate_target_covariances_rbf(3); c.calculate_source_covariances_rbf(3)
    else:
        def resort_knn():
            c.set_target_cloud(tgt)  # invalidates the cached Morton order
            c.find_target_neighbors(20)
        timed("set_cloud+sort (sort kernels, sum)", "sort", resort_knn, reps=max(3, args.reps // 5))
        timed("find_target_neighbors(20)", "knn", lambda: c.find_target_neighbors(20), reps=max(3, args.reps // 5))
        c.find_source_neighbors(20)
        timed("calculate_target_covariances", "cov", lambda: c.calculate_target_covariances(3))
        c.calculate_source_covariances(3)
    timed("calculate_source_covariances_rbf", "rbf", lambda: c.calculate_source_covariances_rbf(3), reps=max(3, args.reps // 5))
    if not args.skip_knn:
        c.calculate_source_covariances(3)
    c.create_target_voxelmap(); c.align()  # a readback gives the engine the voxel-count hint
    timed("create_target_voxelmap", "voxelmap", lambda: c.create_target_voxelmap())
    print("table capacity", c.debug_table_capacity())
    T = np.eye(4)
    timed("update_correspondences (find)", "cost", lambda: c.update_correspondences(T))
    timed("compute_error (H,b)", "cost", lambda: c.compute_error(T, True))
    timed("compute_error (error only)", "cost", lambda: c.compute_error(T, False))
    r = c.align()
    print("align: lin %d err %d launches %d converged %s" % (r["num_linearize"], r["num_error_evals"], r["num_launches"], r["converged"]))
    timed("align (device LM)", "cost", lambda: c.align())
    timed("fitness_score", "fitness", lambda: c.fitness_score(r["T"]), reps=5)
    print("n_corr", c.get_num_correspondences())


if __name__ == "__main__":
    main()
