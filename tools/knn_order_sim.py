"""Would a different DISPATCH ORDER of the one-query waves shorten the 17k k-NN launch? Measured per-query durations (timing build,
FVH_KNN_TIMING) replayed on a model of the chip (8,192 wave slots, waves start in dispatch order as slots free up):
index order (today), tiles by descending box diagonal, tiles by descending measured cost (the unreachable optimum), random tiles."""
import ctypes as C
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess  # noqa: E402

L = capi.load()
tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
for name, cloud in (("source", src), ("target", tgt)):
    c = capi.VGICPCore(0)
    for _ in range(3):
        c.set_source_cloud(cloud); c.find_source_neighbors(20); c.synchronize()
    buf = np.zeros((32768, 8), np.uint64)
    L.fvh_debug_knn_timing(buf.ctypes.data_as(C.c_void_p))
    n = len(cloud)
    t = buf[:n].astype(np.float64)
    dur = (t[:, 4] - t[:, 0]) / 100.0
    order, boxes = c.debug_spatial_order("source")
    ntiles = len(boxes)
    diag = np.linalg.norm(boxes[:, 4:7] - boxes[:, 0:3], axis=1)
    tile_of = np.arange(n) // 64
    tile_cost = np.array([dur[tile_of == k].mean() for k in range(ntiles)])
    tile_max = np.array([dur[tile_of == k].max() for k in range(ntiles)])
    print("%s: %d queries, %d tiles; corr(tile box diagonal, mean duration) = %.2f, corr(diag, max duration) = %.2f" % (
        name, n, ntiles, np.corrcoef(diag, tile_cost)[0, 1], np.corrcoef(diag, tile_max)[0, 1]))

    def span(tile_order, slots=8192):
        q = np.concatenate([np.arange(k * 64, min(n, k * 64 + 64)) for k in tile_order])
        free = [0.0] * slots
        heapq.heapify(free)
        end = 0.0
        for i in q:
            s = heapq.heappop(free)
            e = s + dur[i]
            heapq.heappush(free, e)
            end = max(end, e)
        return end

    rng = np.random.default_rng(0)
    print("  modelled span [us]: index order %.1f | by box diagonal (desc) %.1f | by measured tile cost (desc) %.1f | random %.1f | measured span %.1f" % (
        span(np.arange(ntiles)), span(np.argsort(-diag)), span(np.argsort(-tile_cost)), span(rng.permutation(ntiles)), (t[:, 4].max() - t[:, 0].min()) / 100.0))
    c.close()
