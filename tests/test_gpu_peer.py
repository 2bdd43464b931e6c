"""Multi-GPU path through the ENGINE (not the oracle): ranks of a sharded registration exchanging through peer-mapped regions
(fvh_vgicp_peer_*). The test box has one GPU, so the ranks are (a) separate PROCESSES sharing it -- hipIpc handles, exactly the
one-process-per-GPU deployment except that "xGMI" is the local fabric -- and (b) two handles of ONE process driven by two
threads (process-local pointers). In both cases:
  * every rank ends with covariances BIT-IDENTICAL to the unsharded engine's (tile computed locally, rest all-gathered),
  * err / H / b and the final transform equal the unsharded engine's to 1e-11 (the sums are formed in a different order),
    with equal linearisation / error-evaluation counts, and all ranks hold bit-identical results (rank-order sums, LM step
    replicated, no broadcast)."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unsharded(n_t, n_s, search, cov):
    from fast_gicp_amd import capi, workloads
    tgt, src, T = workloads.synthetic_pair(n_t, n_s, seed=21, extent=40.0)
    c = capi.VGICPCore(0)
    c.set_resolution(0.5); c.set_neighbor_search_method(search); c.set_kernel_params(0.5, 2.5)
    c.set_target_cloud(tgt); c.set_source_cloud(src)
    if cov == "rbf":
        c.calculate_target_covariances_rbf(3); c.calculate_source_covariances_rbf(3)
    else:
        c.find_target_neighbors(20); c.calculate_target_covariances(3)
        c.find_source_neighbors(20); c.calculate_source_covariances(3)
    c.create_target_voxelmap()
    out = dict(cov_t=c.get_covariances("target"), cov_s=c.get_covariances("source"))
    out["e"], out["H"], out["b"] = c.linearize(np.eye(4))
    out["ncorr"] = c.get_num_correspondences()
    out["r"] = c.align()
    out["r2"] = c.align(T)
    c.close()
    return out


def _check(res, ref):
    for r in res:
        assert np.array_equal(r["cov_t"], ref["cov_t"]) and np.array_equal(r["cov_s"], ref["cov_s"]), "all-gathered covariances differ from the unsharded ones"
        assert int(r["ncorr"]) > 0
        assert abs(float(r["e"]) - ref["e"]) <= 1e-11 * abs(ref["e"])
        assert util.rel_err(r["H"], ref["H"]) < 1e-11 and util.rel_err(r["b"], ref["b"]) < 1e-9
        assert bool(r["converged"]) and ref["r"]["converged"]
        assert int(r["nlin"]) == ref["r"]["num_linearize"] and int(r["nerr"]) == ref["r"]["num_error_evals"]
        assert util.rel_err(r["T"], ref["r"]["T"]) < 1e-11 and util.rel_err(r["Hf"], ref["r"]["H"]) < 1e-11
        assert util.rel_err(r["T2"], ref["r2"]["T"]) < 1e-11
    for r in res[1:]:  # replicated LM on rank-order sums: bit-identical across ranks
        assert np.array_equal(r["T"], res[0]["T"]) and np.array_equal(r["Hf"], res[0]["Hf"]) and float(r["e"]) == float(res[0]["e"]) and np.array_equal(r["H"], res[0]["H"])
    assert sum(int(r["ncorr"]) for r in res) == ref["ncorr"]  # the tiles partition the correspondences


@pytest.mark.parametrize("world,n_t,n_s,search,cov", [(2, 60000, 40000, 1, "knn"), (2, 9000, 7000, 0, "rbf"), (3, 20000, 15000, 1, "knn"),
                                                       (8, 30000, 24000, 1, "knn")])  # (8 = FVH_MAX_PEERS, the rank count north_star names: every mailbox slot in use)
def test_sharded_registration_processes_sharing_the_gpu(tmp_path, world, n_t, n_s, search, cov):
    port = _free_port()
    env = dict(os.environ)
    env.setdefault("FVH_SORT_MODE", "1")  # several processes on one GPU: no cooperative sort grids next to the persistent LM grids
    procs = []
    for rank in range(world):
        out = os.path.join(str(tmp_path), "rank%d.npz" % rank)
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.ROOT, "tests", "peer_worker.py"), str(rank), str(world), str(port), out, str(n_t), str(n_s), str(search), cov],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=150 if world <= 3 else 400)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank of the sharded registration hung")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    ref = _unsharded(n_t, n_s, search, cov)
    _check(res, ref)
    # the point of the mailbox: a sharded align is still ONE launch per rank (reported, and required when nothing aborted)
    print("launches per rank:", [int(r["launches"]) for r in res], "aborts:", [int(r["aborts"]) for r in res])
    for r in res:
        if int(r["aborts"]) == 0:
            assert int(r["launches"]) == 1 and int(r["launches2"]) == 1


def test_target_map_sharded_by_tile_and_halo():
    """"target voxel map sharded by the same tiles + 1-voxel halo". Two ranks (processes sharing the GPU), DIRECT7, each building
    only the voxels around T_guess * (its tile of the source) + halo: the registration equals the one on the replicated map to 1e-11
    with equal iteration counts, every rank's shard is smaller than the full map, and no align had to fall back to the full map."""
    n_t, n_s, search, world = 60000, 40000, 1, 2
    port = _free_port()
    outs, procs = [], []
    import tempfile
    d = tempfile.mkdtemp()
    for rank in range(world):
        out = os.path.join(d, "rank%d.npz" % rank)
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.ROOT, "tests", "peer_worker.py"), str(rank), str(world), str(port), out, str(n_t), str(n_s), str(search), "knn", "shardmap"],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [np.load(o) for o in outs]
    ref = _unsharded(n_t, n_s, search, "knn")
    for r in res:
        assert bool(r["is_shard"]) and int(r["fallbacks"]) == 0, (r["is_shard"], r["fallbacks"])
        assert 0 < int(r["nvox"]) < int(r["nvox_full"]), (int(r["nvox"]), int(r["nvox_full"]))
        assert bool(r["converged"]) and int(r["nlin"]) == ref["r"]["num_linearize"] and int(r["nerr"]) == ref["r"]["num_error_evals"]
        assert util.rel_err(r["T"], ref["r"]["T"]) < 1e-11 and util.rel_err(r["Hf"], ref["r"]["H"]) < 1e-11
        assert util.rel_err(r["T2"], ref["r2"]["T"]) < 1e-11
    assert np.array_equal(res[0]["T"], res[1]["T"])
    print("voxels per rank:", [int(r["nvox"]) for r in res], "of", int(res[0]["nvox_full"]))


def test_sharded_registration_two_handles_one_process():
    """Two ranks as two handles of this process (two host threads): the regions are shared by pointer (IPC cannot map one's
    own allocation). Only one handle of a process may run the persistent kernel at a time, so one rank takes the
    one-launch-per-transition route -- the exchange numbering is the same on both routes, so they interoperate."""
    from fast_gicp_amd import capi, workloads
    n_t, n_s, search = 30000, 20000, 1
    tgt, src, T = workloads.synthetic_pair(n_t, n_s, seed=21, extent=40.0)
    cores = [capi.VGICPCore(0) for _ in range(2)]
    exports = [c.peer_export(max(n_t, n_s)) for c in cores]
    for rank, c in enumerate(cores):
        c.set_resolution(0.5); c.set_neighbor_search_method(search)
        c.peer_attach(2, rank, 2, [h for h, _ in exports], [p for _, p in exports])
    res, errs = [None, None], []

    def run(rank):
        try:
            c = cores[rank]
            c.set_target_cloud(tgt); c.set_source_cloud(src)
            c.find_target_neighbors(20); c.calculate_target_covariances(3)
            c.find_source_neighbors(20); c.calculate_source_covariances(3)
            c.create_target_voxelmap()
            e, H, b = c.linearize(np.eye(4))
            ncorr = c.get_num_correspondences()
            r = c.align()
            r2 = c.align(T)
            res[rank] = dict(cov_t=c.get_covariances("target"), cov_s=c.get_covariances("source"), e=e, H=H, b=b, T=r["T"], Hf=r["H"], converged=r["converged"],
                             nlin=r["num_linearize"], nerr=r["num_error_evals"], T2=r2["T"], ncorr=ncorr)
        except Exception as ex:  # noqa: BLE001
            errs.append((rank, repr(ex)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errs, errs
    assert all(r is not None for r in res)
    for c in cores:
        c.peer_detach(); c.close()
    ref = _unsharded(n_t, n_s, search, "knn")
    _check(res, ref)


def test_peer_errors():
    from fast_gicp_amd import capi
    c = capi.VGICPCore(0)
    with pytest.raises(capi.FvhError):
        c.peer_attach(2, 0, 1, [b"\0" * 64] * 2)      # no export yet
    h, p = c.peer_export(1000)
    with pytest.raises(capi.FvhError):
        c.peer_attach(9, 0, 1, [h] * 9)                # more than 8 ranks
    c.peer_attach(1, 0, 1, [h], [p])                   # a single rank is a no-op communicator
    pts = np.random.default_rng(0).uniform(-5, 5, size=(3000, 3)).astype(np.float32)
    c.set_target_cloud(pts); c.find_target_neighbors(20); c.calculate_target_covariances(3); c.create_target_voxelmap()
    c.set_source_cloud(pts); c.find_source_neighbors(20); c.calculate_source_covariances(3)
    assert c.align()["converged"]
    c.peer_detach()
    c.close()


def test_peer_selfcheck():
    """fvh_vgicp_peer_selfcheck: a store of every rank reaches every rank's region (the in-kernel mailboxes' path), checked once
    after attaching. Two handles of one process: both call it -> ok; only one calls it -> FVH_ERR_COMM naming the silent rank,
    and the next (complete) check works again."""
    from fast_gicp_amd import capi
    cores = [capi.VGICPCore(0) for _ in range(2)]
    exports = [c.peer_export(1000) for c in cores]
    for rank, c in enumerate(cores):
        c.peer_attach(2, rank, 2, [h for h, _ in exports], [p for _, p in exports])
    out = [None, None]

    def check(rank, timeout):
        try:
            out[rank] = cores[rank].peer_selfcheck(timeout)
        except capi.FvhError as ex:
            out[rank] = ex

    th = [threading.Thread(target=check, args=(r, 5.0)) for r in range(2)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    assert out == [0, 0], out
    check(0, 0.2)  # rank 1 stays silent
    assert isinstance(out[0], capi.FvhError) and "rank(s) 1" in str(out[0]), out[0]
    # rank 1 catches up with the missed round (its counter must match rank 0's), then a complete round passes
    out[1] = None
    check(1, 0.2)
    th = [threading.Thread(target=check, args=(r, 5.0)) for r in range(2)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    assert out == [0, 0], out
    solo = capi.VGICPCore(0)
    with pytest.raises(capi.FvhError):
        solo.peer_selfcheck()  # nothing attached
    solo.close()
    for c in cores:
        c.peer_detach(); c.close()
