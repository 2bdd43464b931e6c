"""Worker of tests/test_gpu_peer.py: one RANK of a sharded registration through the engine's peer-mapped exchange.
Several of these processes share the one GPU of the test box (the watchdogs turn anything stuck into an error, not a hang).
    python tests/peer_worker.py <rank> <world> <port> <out.npz> <n_target> <n_source> <search> <cov>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    n_t, n_s, search, cov = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
    shard_map = len(sys.argv) > 9 and sys.argv[9] == "shardmap"
    import torch.distributed as dist
    from fast_gicp_amd import capi, distributed as D, workloads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tgt, src, T = workloads.synthetic_pair(n_t, n_s, seed=21, extent=40.0)
    core = capi.VGICPCore(0)
    core.set_resolution(0.5); core.set_neighbor_search_method(search); core.set_kernel_params(0.5, 2.5)
    sh = D.ShardedVGICP(core, rank, world, dist, collective="peer")
    sh.attach_peers(max(n_t, n_s), device_index=0)
    core.set_target_cloud(tgt); core.set_source_cloud(src)
    if cov == "rbf":
        core.calculate_target_covariances_rbf(3); core.calculate_source_covariances_rbf(3)
    else:
        core.find_target_neighbors(20); core.calculate_target_covariances(3)
        core.find_source_neighbors(20); core.calculate_source_covariances(3)
    core.create_target_voxelmap()
    cov_t, cov_s = core.get_covariances("target"), core.get_covariances("source")
    e, H, b = core.linearize(np.eye(4))
    ncorr = core.get_num_correspondences()
    nvox_full = len(core.get_voxelmap()[0])
    if shard_map:  # the target map sharded by the ranks' tiles + halo (fvh_vgicp_set_target_map_sharding): built per align from here on
        core.set_target_map_sharding(True, 2)
    r = core.align()
    shard_state = core.debug_map_shard()      # (before any getter: the voxel getters rebuild the whole map after a sharded align)
    nvox = core.debug_live_map_voxels()
    if shard_map:
        assert len(core.get_voxelmap()[0]) == nvox_full and not core.debug_map_shard()[0]  # ... as the header says
    r2 = core.align(T)  # a second collective align on the same handles (exchange counters carry on)
    np.savez(out, nvox_full=nvox_full, nvox=nvox, is_shard=shard_state[0], fallbacks=core.debug_map_shard()[1], cov_t=cov_t, cov_s=cov_s, e=e, H=H, b=b, T=r["T"], Hf=r["H"], converged=r["converged"], nlin=r["num_linearize"], nerr=r["num_error_evals"],
             launches=r["num_launches"], aborts=core.debug_persist_aborts(), T2=r2["T"], launches2=r2["num_launches"], ncorr=ncorr)
    dist.barrier()
    core.peer_detach()
    core.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
