"""N > 1 path on CPU: spatial-tile partition + the sharded LM loop with a real collective
(torch.distributed gloo, world_size 2, 127.0.0.1).  The per-rank evaluator is the ORACLE on the rank's tile
(test infrastructure standing in for the GPU engine, which cannot run here); the host logic under test is
fast_gicp_amd.distributed (partition, all-reduce packing, replicated LM recursion)."""
import os
import socket

import numpy as np
import pytest

from tests import util


def test_spatial_tile_partition_is_a_compact_disjoint_cover():
    from fast_gicp_amd import distributed as D
    _, src = util.bundled_pair()
    for n in (2, 4, 8):
        tiles = D.spatial_tile_partition(src, n)
        allidx = np.concatenate(tiles)
        assert len(allidx) == len(src) and len(np.unique(allidx)) == len(src)
        assert max(map(len, tiles)) - min(map(len, tiles)) <= 1
        full = np.prod(src.max(0) - src.min(0))
        vols = [np.prod(src[t].max(0) - src[t].min(0)) for t in tiles]
        assert sum(vols) < 1.5 * full and np.median(vols) < full / n * 2.5  # tiles are spatially compact, not random subsets


def test_se3_exp_matches_oracle():
    from fast_gicp_amd import distributed as D
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    for s in (1e-7, 1e-2, 1.0):
        a = rng.normal(size=6) * s
        np.testing.assert_allclose(D.se3_exp(a), O.se3_exp(a), atol=1e-14)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, search, out_dir):
    import torch.distributed as dist
    from fast_gicp_amd import distributed as D
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tgt, src = util.bundled_pair()
    tgt, src = tgt[:6000], src[:6000]
    cov_t, cov_s = O.covariances_knn(tgt, 20, O.PLANE, threads=2), O.covariances_knn(src, 20, O.PLANE, threads=2)
    tile = D.spatial_tile_partition(src, world)[rank]
    g = O.FastVGICP(threads=2, search=search)
    g.set_target(tgt); g.set_source(src[tile])
    g.set_target_covs(cov_t); g.set_source_covs(cov_s[tile])
    g.prepare()

    def allreduce(v):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(v, np.float64))
        dist.all_reduce(t)
        return t.numpy()

    lsq = D.ShardedLsq(lambda T: g.linearize(T), lambda T: g.compute_error(T), allreduce)
    e, H, b = lsq.linearize(np.eye(4))
    r = lsq.align()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), T=r["T"], H=r["H"], converged=r["converged"], e=e, H0=H, b0=b, n=len(tile), it=r["nr_iterations"])
    dist.destroy_process_group()


@pytest.mark.parametrize("search", [2, 0])
def test_sharded_lm_world2_gloo_equals_unsharded(tmp_path, search):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, search, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # all ranks hold the same answer (replicated LM on identical sums, no broadcast)
    assert np.array_equal(res[0]["T"], res[1]["T"]) and res[0]["it"] == res[1]["it"]
    assert res[0]["n"] + res[1]["n"] == 6000
    # and it equals the unsharded registration
    tgt, src = util.bundled_pair()
    tgt, src = tgt[:6000], src[:6000]
    g = O.FastVGICP(threads=2, search=search)
    g.set_target(tgt); g.set_source(src)
    g.set_target_covs(O.covariances_knn(tgt, 20, O.PLANE, threads=2)); g.set_source_covs(O.covariances_knn(src, 20, O.PLANE, threads=2))
    g.prepare()
    e, H, b = g.linearize(np.eye(4))
    assert abs(res[0]["e"] - e) <= 1e-10 * abs(e)
    assert util.rel_err(res[0]["H0"], H) < 1e-10 and util.rel_err(res[0]["b0"], b) < 1e-10
    g2 = O.FastVGICP(threads=2, search=search)
    g2.set_target(tgt); g2.set_source(src)
    ro = g2.align()
    assert bool(res[0]["converged"]) and ro["converged"]
    assert util.rel_err(res[0]["T"], ro["T"]) < 1e-8
    assert util.rel_err(res[0]["H"], ro["H"]) < 1e-8
