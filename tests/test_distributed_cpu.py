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


@pytest.mark.parametrize("search, world", [(2, 2), (0, 2), (0, 4), (0, 8)])
def test_sharded_lm_world2_gloo_equals_unsharded(tmp_path, search, world):
    """(world 4 and 8: round 6 -- the partition, the packing of the all-reduced sums and the replicated LM recursion at the rank count
    north_star names)"""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    port = _free_port()
    mp.spawn(_worker, args=(world, port, search, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # all ranks hold the same answer (replicated LM on identical sums, no broadcast)
    assert all(np.array_equal(res[0]["T"], r["T"]) and res[0]["it"] == r["it"] for r in res[1:])
    assert sum(int(r["n"]) for r in res) == 6000
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


class _FakeRcclCore:
    """Stands in for capi.VGICPCore on a box without GPUs, for the route collective="rccl": what the ENGINE does with a communicator
    attached (fvh_vgicp_comm_init: internal sharding) is restated here on the ORACLE with gloo collectives -- neighbour search / covariances
    for the rank's spatial tile against the FULL cloud, an all-gather of the tiles' covariances (ncclAllGather in the engine), the cost
    evaluation over the rank's tile with an all-reduce of the sums (ncclAllReduce) -- so that two CPU ranks can check the host logic of
    fast_gicp_amd.distributed (id hand-over, same calls on every rank) AND that this decomposition equals the unsharded registration."""

    def __init__(self, dist, search):
        from oracle import oracle as O
        self.O, self.dist, self.search = O, dist, search
        self.calls, self.comm = [], None
        self.src = self.tgt = self.src_cov = self.tgt_cov = None
        self.tile = None

    def _tile(self, xyz):
        from fast_gicp_amd import distributed as D
        _, nranks, rank = self.comm
        return D.spatial_tile_partition(xyz, nranks)[rank]

    def _sharded_covariances(self, xyz, k, reg):
        tile = self._tile(xyz)
        mine = self.O.covariances_knn(xyz, k, reg, threads=2)[tile]  # queries of the tile, candidates = the whole cloud (an exact, implicit halo)
        parts = [None] * self.comm[1]
        self.dist.all_gather_object(parts, (tile, mine))              # ncclAllGather of the 32 B / point covariances
        full = np.zeros((len(xyz), 3, 3))
        for t, c in parts:
            full[t] = c
        return tile, full

    def set_target_cloud(self, xyz): self.tgt = np.asarray(xyz, np.float32); self.calls.append(("set_target_cloud", len(xyz)))
    def find_target_neighbors(self, k): self.k_t = k
    def calculate_target_covariances(self, reg): _, self.tgt_cov = self._sharded_covariances(self.tgt, self.k_t, reg)
    def create_target_voxelmap(self): self.calls.append(("create_target_voxelmap",))
    def set_source_cloud(self, xyz): self.src = np.asarray(xyz, np.float32); self.src_cov = None; self.calls.append(("set_source_cloud", len(xyz)))
    def find_source_neighbors(self, k): self.k_s = k
    def calculate_source_covariances(self, reg): self.tile, self.src_cov = self._sharded_covariances(self.src, self.k_s, reg)
    def comm_init(self, uid, nranks, rank): self.comm = (bytes(uid), nranks, rank)

    def align(self, guess=None, **lm):
        import torch
        from fast_gicp_amd import distributed as D
        assert self.comm is not None and len(self.src_cov) == len(self.src)
        g = self.O.FastVGICP(threads=2, search=self.search)
        g.set_target(self.tgt); g.set_source(self.src[self.tile])  # the cost evaluation walks this rank's tile of the source; the target map is replicated
        g.set_target_covs(self.tgt_cov); g.set_source_covs(self.src_cov[self.tile])
        g.prepare()

        def allreduce(v):  # ncclAllReduce(sum) of the engine, on gloo
            t = torch.from_numpy(np.ascontiguousarray(v, np.float64))
            self.dist.all_reduce(t)
            return t.numpy()
        r = D.ShardedLsq(lambda T: g.linearize(T), lambda T: g.compute_error(T), allreduce).align(guess)
        r["num_linearize"] = r["num_error_evals"] = 0
        return r


def _rccl_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from fast_gicp_amd import distributed as D
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tgt, src = util.bundled_pair()
    tgt, src = tgt[:6000], src[:6000]
    core = _FakeRcclCore(dist, O.DIRECT7)
    sh = D.ShardedVGICP(core, rank, world, dist)          # collective=None -> "rccl", the documented default
    assert sh.collective == "rccl"
    with pytest.raises(RuntimeError):
        sh.align()                                          # no communicator yet
    uid = [os.urandom(128) if rank == 0 else None]          # (capi.comm_unique_id() on a GPU box)
    dist.broadcast_object_list(uid, src=0)
    sh.init_device_collective(uid[0])
    sh.set_target(tgt)
    sh.set_source(src)
    full_cov = O.covariances_knn(src, 20, O.PLANE, threads=2)
    r = sh.align()
    assert sh.tile is None and [c[0] for c in core.calls] == ["set_target_cloud", "create_target_voxelmap", "set_source_cloud"]  # same calls on the full clouds: the sharding is the engine's
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), T=r["T"], converged=r["converged"], tile=core.tile, uid=np.frombuffer(core.comm[0], np.uint8),
             cov_ok=np.abs(core.src_cov - full_cov).max(), n_src=len(core.src))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_route_host_logic_world2_gloo(tmp_path, world):
    """The route collective="rccl" on two / four gloo ranks: id broadcast -> comm_init on every rank -> the same calls on the same full clouds; the
    engine-side decomposition (tile queries against the full cloud, all-gathered covariances, per-tile cost + all-reduce), restated on the
    oracle, equals the unsharded registration."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    port = _free_port()
    mp.spawn(_rccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert all(np.array_equal(res[0]["uid"], r["uid"]) for r in res[1:])     # the same communicator id reached every rank
    tiles = [set(r["tile"].tolist()) for r in res]
    assert sum(len(t) for t in tiles) == 6000 and len(set().union(*tiles)) == 6000 and all(int(r["n_src"]) == 6000 for r in res)  # every rank holds the full cloud, walks its tile; the tiles are a partition
    assert all(float(r["cov_ok"]) < 1e-12 for r in res)                      # the all-gathered covariances are the full cloud's
    assert all(np.array_equal(res[0]["T"], r["T"]) for r in res[1:]) and bool(res[0]["converged"])
    tgt, src = util.bundled_pair()
    g = O.FastVGICP(threads=2, search=O.DIRECT7)
    g.set_target(tgt[:6000]); g.set_source(src[:6000])
    assert util.rel_err(res[0]["T"], g.align()["T"]) < 1e-8


class _FakeNdtCore:
    """Stands in for capi.NDTCore (P2D) on a box without GPUs: what the ENGINE does with a source tile set (fvh_ndt_set_source_tile: full
    clouds on every rank, the target voxel map replicated, the cost evaluated over the rank's chunk of the source points' Morton order) is
    restated on the ORACLE's NDT, so that two gloo ranks can check ShardedNDT's host logic and that the decomposition equals the
    unsharded registration."""

    def __init__(self):
        from oracle import oracle as O
        self.O, self.tile_of, self.calls = O, None, []
        self.tgt = self.src = None

    def set_source_tile(self, rank, nranks): self.tile_of = (rank, nranks); self.calls.append(("set_source_tile", rank, nranks))
    def set_target_cloud(self, xyz): self.tgt = np.asarray(xyz, np.float32); self.calls.append(("set_target_cloud", len(xyz)))
    def set_source_cloud(self, xyz): self.src = np.asarray(xyz, np.float32); self.calls.append(("set_source_cloud", len(xyz)))

    def create_voxelmaps(self):
        from fast_gicp_amd import distributed as D
        rank, nranks = self.tile_of
        self.tile = D.spatial_tile_partition(self.src, nranks)[rank]
        self.g = self.O.NDT(threads=2, mode=self.O.P2D, search=self.O.DIRECT7)
        self.g.set_target(self.tgt); self.g.set_source(self.src[self.tile])  # the rank's tile of the source against the replicated target map
        self.g.prepare()

    def linearize(self, T): return self.g.linearize(T)
    def compute_error(self, T, derivatives=True): return self.g.compute_error(T)


def _ndt_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from fast_gicp_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tgt, src = util.bundled_pair(leaf=0.25)
    core = _FakeNdtCore()
    sh = D.ShardedNDT(core, rank, world, dist, collective="host")
    sh.set_target(tgt); sh.set_source(src)
    r = sh.align()
    assert core.calls[0] == ("set_source_tile", rank, world) and [c[0] for c in core.calls[1:]] == ["set_target_cloud", "set_source_cloud"]  # full clouds on every rank
    assert core.calls[2][1] == len(src)
    np.savez(os.path.join(out_dir, "ndt_rank%d.npz" % rank), T=r["T"], converged=r["converged"], tile=core.tile)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_ndt_host_route_world2_gloo(tmp_path, world):
    """ShardedNDT(collective="host") on two / four gloo ranks: the source sharded by spatial tile (P2D: points in Morton order), the target map
    replicated, one all-reduce of the normal equations per evaluation -- equals the unsharded NDT registration of the oracle."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    port = _free_port()
    mp.spawn(_ndt_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "ndt_rank%d.npz" % r)) for r in range(world)]
    tgt, src = util.bundled_pair(leaf=0.25)
    tiles = [set(r["tile"].tolist()) for r in res]
    assert sum(len(t) for t in tiles) == len(src) and len(set().union(*tiles)) == len(src)
    assert all(np.array_equal(res[0]["T"], r["T"]) for r in res[1:]) and bool(res[0]["converged"])  # identical sums on every rank: lock-step without a broadcast
    g = O.NDT(threads=2, mode=O.P2D, search=O.DIRECT7)
    g.set_target(tgt); g.set_source(src)
    r = g.align()
    assert r["converged"] and util.rel_err(res[0]["T"], r["T"]) < 1e-8
