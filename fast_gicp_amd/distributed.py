"""Multi-GPU registration: one process per GPU, clouds sharded by spatial tile (= a range of the Morton order), the
32-double normal-equation block {err, b(6), H(21), ...} summed over the ranks once per cost evaluation.

The reference has no multi-GPU path ("Collectives: none"); this follows BASELINE.json's north_star.
Every correspondence contributes independently to (err, H, b), so the only coupling between shards is the sum.

Three ways to run it (ShardedVGICP(collective=...)):

  "peer"    the engine shards INTERNALLY (include/fast_vgicp_hip.h, fvh_vgicp_peer_*): every rank makes the
            same calls on the same full clouds; k-NN / covariance estimation run on the rank's tile and are all-gathered
            through peer-mapped staging areas; the cost evaluation walks the rank's tile and the sums meet in peer-mapped
            mailboxes INSIDE the cost kernel (one xGMI hop, 1.4 us measured) -- a sharded align stays ONE persistent
            launch per rank, no RCCL, no numpy round trip.  This module only carries the 64-byte IPC handles between
            the ranks (torch.distributed all_gather_object) -- see attach_peers().
  "rccl"    (the DEFAULT: `collective=None`) the engine shards internally here too (fvh_vgicp_comm_init): k-NN / covariance estimation on
            the rank's tile + ncclAllGather of the 32 B / point covariances, ncclAllReduce(32 x f64) on the engine stream between the
            launches of the multi-launch LM route. It stays the default until the peer path has run on distinct GPUs over xGMI (so far
            it was only exercised with all ranks on one device).
  "host"    the host-driven `ShardedLsq` below through a caller-supplied all-reduce (torch.distributed gloo on CPU in the
            tests): what the world_size = 2 CPU tests exercise.

The target Gaussian voxel map is replicated on every GPU (64 B per bucket + 16 B of keys -- a 1M-point map is ~130 MB of
288 GB), so DIRECT7/27 lookups need no halo exchange.
"""
import numpy as np


def morton_order(xyz, bits=10):
    """Permutation that sorts points along a 3*bits-bit Morton curve over the cloud's bounding cube."""
    p = np.asarray(xyz, np.float64)
    lo = p.min(axis=0)
    extent = max(float((p.max(axis=0) - lo).max()), 1e-9)
    q = np.minimum(((p - lo) * ((2 ** bits - 1e-3) / extent)).astype(np.uint64), 2 ** bits - 1)
    key = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for a in range(3):
            key |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(key, kind="stable")


def spatial_tile_partition(xyz, nranks):
    """List of index arrays, one per rank: contiguous, equally sized ranges of the Morton order (spatial tiles)."""
    order = morton_order(xyz)
    bounds = np.linspace(0, len(order), nranks + 1).astype(np.int64)
    return [order[bounds[r]:bounds[r + 1]] for r in range(nranks)]


def se3_exp(a):
    """fast_gicp::se3_exp (so3.hpp:80-104), numpy."""
    a = np.asarray(a, np.float64)
    w, v = a[:3], a[3:]
    th2 = float(w @ w)
    if th2 < 1e-10:
        imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0
        real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0
    else:
        th = np.sqrt(th2)
        imag, real = np.sin(0.5 * th) / th, np.cos(0.5 * th)
    qw, (qx, qy, qz) = real, imag * w
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    th = np.sqrt(th2)
    if th < 1e-10:
        V = R
    else:
        O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        V = np.eye(3) + (1 - np.cos(th)) / th2 * O + (th - np.sin(th)) / (th2 * th) * (O @ O)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


class ShardedLsq:
    """LsqRegistration::computeTransformation / step_lm (lsq_registration_impl.hpp:53-168) where every cost
    evaluation is the all-reduced sum of the ranks' local evaluations.

    local_linearize(T) -> (err, H 6x6, b 6) and local_error(T) -> err evaluate THIS rank's shard;
    allreduce(vec) sums a float64 vector over the ranks in place / returns the sum.  All ranks run the same
    (deterministic) LM recursion on identical sums, so they stay in lock-step without any broadcast."""

    def __init__(self, local_linearize, local_error, allreduce, max_iterations=64, rotation_epsilon=2e-3, transformation_epsilon=5e-4, lm_max_iterations=10,
                 lm_init_lambda_factor=1e-9):
        self.local_linearize, self.local_error, self.allreduce = local_linearize, local_error, allreduce
        self.max_iterations, self.rotation_epsilon, self.transformation_epsilon = max_iterations, rotation_epsilon, transformation_epsilon
        self.lm_max_iterations, self.lm_init_lambda_factor = lm_max_iterations, lm_init_lambda_factor
        self.num_allreduce = 0

    def linearize(self, T):
        e, H, b = self.local_linearize(T)
        v = np.concatenate([[e], np.asarray(b, np.float64), np.asarray(H, np.float64).reshape(-1)])
        v = np.asarray(self.allreduce(v), np.float64)
        self.num_allreduce += 1
        return float(v[0]), v[7:].reshape(6, 6), v[1:7]

    def compute_error(self, T):
        v = np.asarray(self.allreduce(np.array([self.local_error(T)], np.float64)), np.float64)
        self.num_allreduce += 1
        return float(v[0])

    def _is_converged(self, delta):
        r = np.abs(delta[:3, :3] - np.eye(3)).max() / self.rotation_epsilon
        t = np.abs(delta[:3, 3]).max() / self.transformation_epsilon
        return max(r, t) < 1

    def align(self, guess=None):
        x0 = np.eye(4) if guess is None else np.asarray(guess, np.float64).copy()
        lam, converged, iters, final_H = -1.0, False, 0, np.eye(6)
        for i in range(self.max_iterations):
            if converged:
                break
            iters = i
            y0, H, b = self.linearize(x0)
            if lam < 0.0:
                lam = self.lm_init_lambda_factor * np.abs(np.diag(H)).max()
            nu, ok, delta = 2.0, False, np.eye(4)
            for _ in range(self.lm_max_iterations):
                d = np.linalg.solve(H + lam * np.eye(6), -b)
                delta = se3_exp(d)
                xi = delta @ x0
                yi = self.compute_error(xi)
                rho = (y0 - yi) / (d @ (lam * d - b))
                if rho < 0:
                    if self._is_converged(delta):
                        ok = True
                        break
                    lam, nu = nu * lam, 2 * nu
                    continue
                x0, final_H, ok = xi, H, True
                lam = lam * max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                break
            if not ok:
                break
            converged = self._is_converged(delta)
        return dict(T=x0, H=final_H, converged=converged, nr_iterations=iters)


class ShardedVGICP:
    """VGICP over the ranks of a torch.distributed process group (one GPU each); see the module docstring for the three
    collectives. `device_collective` (bool) is round 1's spelling: True = "rccl", False = "host". The target voxel map is replicated by
    default; core.set_target_map_sharding(True, margin) shards it too, by the ranks' tiles + halo (peer and rccl routes)."""

    def __init__(self, core, rank, world_size, dist=None, device_collective=None, collective=None):
        self.core, self.rank, self.world_size, self.dist = core, rank, world_size, dist
        if collective is None:
            collective = "rccl" if (device_collective is None or device_collective) else "host"
        assert collective in ("peer", "rccl", "host")
        self.collective = collective
        self.device_collective = collective != "host"
        self._comm_ready = False

    def init_device_collective(self, unique_id_bytes):
        """collective "rccl": the 128-byte RCCL id created on rank 0 (capi.comm_unique_id()) and broadcast to all ranks."""
        assert self.collective == "rccl"
        self.core.comm_init(unique_id_bytes, self.world_size, self.rank)
        self._comm_ready = True

    def attach_peers(self, max_points, device_index=None, selfcheck_timeout=5.0):
        """collective "peer": export this rank's exchange region, swap the IPC handles with all ranks, map theirs.
        `device_index`: the GPU this rank runs on (ranks sharing a GPU share its co-resident workgroup slots); by default the
        device the handle was created on."""
        import os
        assert self.collective == "peer"
        if device_index is None:
            device_index = getattr(self.core, "device", None)
            if device_index is None:
                raise ValueError("attach_peers: pass device_index (the GPU this rank's handle lives on)")
        handle, ptr = self.core.peer_export(int(max_points))
        mine = (handle, ptr, os.getpid(), int(device_index))
        if self.world_size > 1:
            if self.dist is None:
                raise RuntimeError("attach_peers needs a torch.distributed process group to carry the IPC handles")
            allv = [None] * self.world_size
            self.dist.all_gather_object(allv, mine)
        else:
            allv = [mine]
        same_dev = sum(1 for v in allv if v[3] == int(device_index))
        local = [v[1] if (v[2] == os.getpid()) else 0 for v in allv]
        self.core.peer_attach(self.world_size, self.rank, same_dev, [v[0] for v in allv], local)
        if self.dist is not None and self.world_size > 1:
            self.dist.barrier()  # every rank has mapped every region before anybody writes into one
        if self.world_size > 1:
            self.core.peer_selfcheck(selfcheck_timeout)  # a store of every rank reaches every rank (xGMI / IPC): a clear error now, not a time-out inside a registration
        self._comm_ready = True

    def collective_description(self):
        return {"peer": "peer-mapped mailboxes inside the persistent LM kernel (32 x f64 + tag per rank and evaluation, rank-order sum); covariances all-gathered through peer-mapped staging",
                "rccl": "ncclAllReduce(32 x f64) on the engine stream, once per cost evaluation",
                "host": "host all-reduce (torch.distributed) of 43 doubles per evaluation"}[self.collective]

    def set_target(self, xyz, k=20, regularization=3):
        self.core.set_target_cloud(xyz)
        self.core.find_target_neighbors(k)          # collective "peer": this rank's tile, then the all-gather inside calculate_*
        self.core.calculate_target_covariances(regularization)
        self.core.create_target_voxelmap()

    def set_source(self, full_xyz, k=20, regularization=3):
        full_xyz = np.ascontiguousarray(full_xyz, np.float32)
        if self.collective in ("peer", "rccl"):
            # the engine shards internally on both device routes: same calls on the same full cloud on every rank, no host round trip -- k-NN
            # and covariances on the rank's tile, all-gathered through peer-mapped staging ("peer") or ncclAllGather ("rccl")
            self.core.set_source_cloud(full_xyz)
            self.core.find_source_neighbors(k)
            self.core.calculate_source_covariances(regularization)
            self.tile = None
            return
        # "host": covariances need neighbours across tile borders, so they are computed on the full cloud and the rank then keeps only
        # its tile for the cost evaluations
        tile = spatial_tile_partition(full_xyz, self.world_size)[self.rank]
        self.core.set_source_cloud(full_xyz)
        self.core.find_source_neighbors(k)
        self.core.calculate_source_covariances(regularization)
        covs = self.core.get_covariances("source").astype(np.float64)
        self.core.set_source_cloud(full_xyz[tile])
        self.core.set_source_covariances(covs[tile])
        self.tile = tile

    def align(self, guess=None, **lm):
        if self.device_collective:
            if not self._comm_ready:
                raise RuntimeError("call attach_peers() / init_device_collective() first")
            return self.core.align(guess, **lm)
        import torch

        def allreduce(v):
            t = torch.from_numpy(np.ascontiguousarray(v, np.float64))
            if self.dist is not None and self.world_size > 1:
                self.dist.all_reduce(t)
            return t.numpy()

        lsq = ShardedLsq(lambda T: self.core.linearize(T), lambda T: self.core.compute_error(T, derivatives=False), allreduce, **{
            "max_iterations": lm.get("max_iterations", 64), "rotation_epsilon": lm.get("rotation_epsilon", 2e-3),
            "transformation_epsilon": lm.get("transformation_epsilon", 5e-4), "lm_max_iterations": lm.get("lm_max_iterations", 10),
            "lm_init_lambda_factor": lm.get("lm_init_lambda_factor", 1e-9)})
        return lsq.align(guess)


class ShardedNDT:
    """NDT (P2D or D2D) over the ranks of a process group, sharded by spatial tile of the SOURCE (fvh_ndt_set_source_tile): every rank holds
    the full clouds -- the target voxel map is replicated --, P2D walks the rank's chunk of the source points' Morton order, D2D the rank's
    chunk of the source map's voxels ranked by voxel key (a canonical order: the map's own voxel list is ordered by atomics and differs
    between two builds of the same map). Routes:
      "rccl"  fvh_ndt_comm_init: ncclAllReduce(32 x f64) on the engine stream after every cost launch of the multi-launch LM loop;
      "host"  the host-driven ShardedLsq above through torch.distributed (gloo on CPU in the tests): one all-reduce of 43 doubles per evaluation.
    (There is no peer-mailbox route for NDT handles.)"""

    def __init__(self, core, rank, world_size, dist=None, collective="rccl"):
        assert collective in ("rccl", "host")
        self.core, self.rank, self.world_size, self.dist, self.collective = core, rank, world_size, dist, collective
        self._comm_ready = False
        core.set_source_tile(rank, world_size)

    def init_device_collective(self, unique_id_bytes):
        assert self.collective == "rccl"
        self.core.comm_init(unique_id_bytes, self.world_size, self.rank)
        self._comm_ready = True

    def collective_description(self):
        return {"rccl": "ncclAllReduce(32 x f64) on the engine stream, once per cost evaluation", "host": "host all-reduce (torch.distributed) of 43 doubles per evaluation"}[self.collective]

    def set_target(self, xyz):
        self.core.set_target_cloud(xyz)

    def set_source(self, full_xyz):
        self.core.set_source_cloud(np.ascontiguousarray(full_xyz, np.float32))  # the FULL cloud on every rank: the engine cuts the tile

    def align(self, guess=None, **lm):
        if self.collective == "rccl":
            if not self._comm_ready:
                raise RuntimeError("call init_device_collective() first")
            return self.core.align(guess, **lm)
        import torch
        self.core.create_voxelmaps()

        def allreduce(v):
            t = torch.from_numpy(np.ascontiguousarray(v, np.float64))
            if self.dist is not None and self.world_size > 1:
                self.dist.all_reduce(t)
            return t.numpy()

        lsq = ShardedLsq(lambda T: self.core.linearize(T), lambda T: self.core.compute_error(T, derivatives=False), allreduce, **{
            "max_iterations": lm.get("max_iterations", 64), "rotation_epsilon": lm.get("rotation_epsilon", 2e-3),
            "transformation_epsilon": lm.get("transformation_epsilon", 5e-4), "lm_max_iterations": lm.get("lm_max_iterations", 10),
            "lm_init_lambda_factor": lm.get("lm_init_lambda_factor", 1e-9)})
        return lsq.align(guess)
