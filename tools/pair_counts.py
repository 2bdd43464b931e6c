#!/usr/bin/env python
"""What the culled exact searches actually evaluate (VERDICT r4 #2b: the flop fraction of the VALU-bound stages). Needs the debug build of
the library (`python tools/build_variants.py knn_timing="-DFVH_KNN_TIMING"`, selected with FVH_LIB_PATH): its k-NN and RBF kernels count
candidate points and box tests per launch.
    FVH_LIB_PATH=fast_gicp_amd/lib/variants/knn_timing/libfast_vgicp_hip.so python tools/pair_counts.py out.json
Per workload of bench.py: candidate point distances (8 flop each: 3 sub, 3 mul / fma, 2 add; the RBF sweep adds an exp and ~22 flop of
weighted moments per candidate, counted as 30) and box tests (12 flop each) of ONE launch of the stage's kernel on the source cloud."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fast_gicp_amd import capi  # noqa: E402


def counts(reset=True):
    buf = (C.c_ulonglong * 8)()
    rc = capi.load().fvh_debug_pair_counts(buf, 1 if reset else 0)
    if rc != 0:
        raise SystemExit("fvh_debug_pair_counts failed (%d): is FVH_LIB_PATH the -DFVH_KNN_TIMING build?" % rc)
    return np.array(list(buf), np.float64)


def main(out):
    if not hasattr(capi.load(), "fvh_debug_pair_counts"):
        raise SystemExit("this library has no fvh_debug_pair_counts: build the knn_timing variant and point FVH_LIB_PATH at it")
    res = {}
    for key, workload, cov in (("bundled17k_knn", "bundled17k", "knn"), ("synth1m_knn", "synth1m", "knn"), ("synth100k_rbf_rbf", "synth100k", "rbf"), ("bundled17k_rbf_rbf", "bundled17k", "rbf")):
        tgt, src, res_, desc = bench.make_workload(workload)
        c = capi.VGICPCore(0)
        c.set_kernel_params(0.5, 2.5)
        c.set_source_cloud(src)
        c.find_source_neighbors(20); c.synchronize()  # (sorts the cloud)
        counts(True)
        if cov == "knn":
            c.find_source_neighbors(20)
        else:
            c.calculate_source_covariances_rbf(capi.REG_PLANE)
        c.synchronize()
        v = counts(True)
        pts, boxes, q = (v[0], v[1], v[4]) if cov == "knn" else (v[2], v[3], v[5])
        flop_pt = 8.0 if cov == "knn" else 30.0
        res[key] = {"queries": int(q), "candidate_points_per_launch": int(pts), "box_tests_per_launch": int(boxes), "candidates_per_query": round(pts / max(q, 1), 1),
                    "flop_per_launch": float(pts * flop_pt + boxes * 12.0), "flop_per_candidate": flop_pt, "flop_per_box_test": 12.0,
                    "full_sweep_candidates": float(len(src)) ** 2}
        c.close()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
