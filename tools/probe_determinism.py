"""Debug probe: are repeated aligns of one handle, and aligns of different handles, bit-identical (same grid)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess
tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
def make(search):
    c = capi.VGICPCore(0)
    c.set_neighbor_search_method(search)
    c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    return c
for search in (0, 1, 2):
    a = make(search)
    rs = [a.align() for _ in range(40)]
    b = make(search)
    rb = [b.align() for _ in range(5)]
    T0 = rs[0]["T"]
    diff = [i for i, r in enumerate(rs) if not np.array_equal(r["T"], T0)]
    diffb = [i for i, r in enumerate(rb) if not np.array_equal(r["T"], T0)]
    print("search", search, "grid", a.debug_persist_grid(), "launches", sorted(set(r["num_launches"] for r in rs)), "calls differing from call 0:", diff, "other handle:", diffb,
          "max |dT|", max(np.abs(r["T"] - T0).max() for r in rs + rb), "evals", sorted(set((r["num_linearize"], r["num_error_evals"]) for r in rs + rb)))
    a.close(); b.close()
