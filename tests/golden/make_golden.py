#!/usr/bin/env python
"""Generates tests/golden/bundled_pair.json from the ORACLE (oracle/liboracle.so) on the bundled scans.

The reference itself cannot be run here (no PCL/Eigen), so these are the oracle's answers, frozen: any later change
to the oracle or the engine that moves them is caught. Values that the reference publishes or that SURVEY.md recorded
independently (point counts, fitness, translations) are asserted at generation time.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import util  # noqa: E402


def main():
    out = {"source": "oracle/liboracle.so (fp64 restatement of FastVGICP / NDTCuda formulas)", "cases": {}}
    tgt, src = util.bundled_pair()
    assert (len(tgt), len(src)) == (17047, 17334)
    out["counts_head_preprocessing"] = [len(tgt), len(src)]
    t2, s2 = util.bundled_pair(origin_filter=False)
    assert (len(t2), len(s2)) == (17249, 17518)  # README.md:116
    out["counts_readme_preprocessing"] = [len(t2), len(s2)]
    poses = {"identity": np.eye(4), "relative_txt": util.relative_pose()}
    for name, search in (("vgicp_direct1", O.DIRECT1), ("vgicp_direct7", O.DIRECT7), ("vgicp_direct27", O.DIRECT27)):
        g = O.FastVGICP(search=search)
        g.set_target(tgt); g.set_source(src)
        r = g.align()
        case = {"T": r["T"].tolist(), "H": r["H"].tolist(), "converged": r["converged"], "iterations": r["iterations"], "num_linearize": r["num_linearize"],
                "num_error_evals": r["num_error_evals"], "fitness": g.fitness(), "num_voxels": int(len(g.get_voxelmap()[0])), "linearize": {}}
        h = O.FastVGICP(search=search)
        h.set_target(tgt); h.set_source(src); h.prepare()
        for pn, T in poses.items():
            e, H, b = h.linearize(T)
            case["linearize"][pn] = {"error": e, "H": H.tolist(), "b": b.tolist(), "num_correspondences": h.num_correspondences()}
        out["cases"][name] = case
    for name, max_dist in (("gicp", None), ("gicp_maxdist1", 1.0)):  # FastGICP (nearest target point), fast_gicp_impl.hpp:118-240
        g = O.FastVGICP()
        g.set_gicp_mode(True, 3.4028234663852886e38 if max_dist is None else max_dist)
        g.set_target(tgt); g.set_source(src)
        r = g.align()
        case = {"T": r["T"].tolist(), "H": r["H"].tolist(), "converged": r["converged"], "iterations": r["iterations"], "fitness": g.fitness(), "max_correspondence_distance": max_dist,
                "linearize": {}}
        h = O.FastVGICP()
        h.set_gicp_mode(True, 3.4028234663852886e38 if max_dist is None else max_dist)
        h.set_target(tgt); h.set_source(src); h.prepare()
        for pn, T in poses.items():
            e, H, b = h.linearize(T)
            case["linearize"][pn] = {"error": e, "H": H.tolist(), "b": b.tolist(), "num_correspondences": h.num_correspondences()}
        out["cases"][name] = case
    for name, mode in (("ndt_d2d", O.D2D), ("ndt_p2d", O.P2D)):
        g = O.NDT(mode=mode)
        g.set_target(tgt); g.set_source(src)
        r = g.align()
        out["cases"][name] = {"T": r["T"].tolist(), "converged": r["converged"], "iterations": r["iterations"], "fitness": g.fitness()}
    # cross-checks against numbers recorded independently in SURVEY.md 8(c)(4)
    np.testing.assert_allclose(np.array(out["cases"]["vgicp_direct1"]["T"])[:3, 3], [0.498359, 0.117208, -0.029736], atol=2e-6)
    assert abs(out["cases"]["vgicp_direct1"]["fitness"] - 0.205022) < 2e-6
    np.testing.assert_allclose(np.array(out["cases"]["vgicp_direct27"]["T"])[:3, 3], [0.503962, 0.074634, -0.025146], atol=2e-6)
    assert abs(out["cases"]["vgicp_direct27"]["fitness"] - 0.198792) < 2e-6 and out["cases"]["vgicp_direct27"]["num_voxels"] == 1087
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "bundled_pair.json"), "w"), indent=1)
    print("wrote tests/golden/bundled_pair.json")


if __name__ == "__main__":
    main()
