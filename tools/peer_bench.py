#!/usr/bin/env python
"""Sharded-registration timing with N ranks (processes) on whatever GPUs the box has -- on the 1-GPU authoring box all ranks
share GPU 0, which measures the OVERHEAD of the peer-mapped exchange path (no speed-up is possible there), on an 8-GPU node
`--devices 0,1,...` gives the real scaling of one registration.  BASELINE configs[4]: 1M-point map <-> 100k-point scan.
    python tools/peer_bench.py --ranks 2 [--devices 0,0] [--map 1000000] [--scan 100000] [--steps 20]"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a):
    import numpy as np
    import torch.distributed as dist
    from fast_gicp_amd import capi, distributed as D, workloads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.ranks)
    devs = [int(d) for d in a.devices.split(",")]
    dev = devs[a.rank % len(devs)]
    tgt, src, _ = workloads.synthetic_pair(a.map, a.scan, seed=44, extent=150.0)
    core = capi.VGICPCore(dev)
    core.set_resolution(0.5); core.set_neighbor_search_method(capi.DIRECT7)

    def step():
        core.set_source_cloud(src); core.find_source_neighbors(20); core.calculate_source_covariances(capi.REG_PLANE)
        return core.align()

    def timed(label):
        step(); step()
        core.profile_reset(); core.profile_enable(True)
        step()
        core.profile_enable(False)
        stages = {c: round(core.profile_get(c)[0] * 1e3, 1) for c in ("sort", "knn", "cov", "peer_gather", "cost")}
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r = step()
        core.synchronize()
        dist.barrier()
        return (time.perf_counter() - t0) / a.steps * 1e3, r, stages

    res = {"rank": a.rank, "device": dev}
    if a.rank == 0 or not a.single_on_rank0_only:
        core.set_target_cloud(tgt); core.find_target_neighbors(20); core.calculate_target_covariances(capi.REG_PLANE); core.create_target_voxelmap()
    if a.single_on_rank0_only:  # on a shared GPU the unsharded reference must not compete with copies of itself
        if a.rank == 0:
            step(); step()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                r1 = step()
            core.synchronize()
            res["single_ms"] = (time.perf_counter() - t0) / a.steps * 1e3
            res["T_single"] = r1["T"].tolist()
            res["aborts_single_phase"] = core.debug_persist_aborts()
        dist.barrier()
    sh = D.ShardedVGICP(core, a.rank, a.ranks, dist, collective="peer")
    sh.attach_peers(a.map, device_index=dev)
    sh.set_target(tgt)
    ms, r, stages = timed("sharded")
    res.update(sharded_ms=ms, launches=r["num_launches"], evals=r["num_linearize"] + r["num_error_evals"], aborts=core.debug_persist_aborts(), converged=bool(r["converged"]),
               stages_us=stages, T=r["T"].tolist())
    dist.barrier()
    core.peer_detach(); core.close()
    print("PEER_BENCH " + json.dumps(res), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--devices", default="0")
    ap.add_argument("--map", type=int, default=1_000_000)
    ap.add_argument("--scan", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--single-on-rank0-only", action="store_true", default=True)
    a = ap.parse_args()
    if a.rank >= 0:
        return worker(a)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    if len(set(a.devices.split(","))) < a.ranks:
        env.setdefault("FVH_SORT_MODE", "1")  # ranks share a GPU: no cooperative sort grids next to the persistent LM grids
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ranks", str(a.ranks), "--devices", a.devices, "--map", str(a.map), "--scan", str(a.scan), "--steps", str(a.steps),
                               "--rank", str(r), "--port", str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(a.ranks)]
    import numpy as np
    out = []
    for p in procs:
        log = p.communicate(timeout=600)[0]
        lines = [l for l in log.splitlines() if l.startswith("PEER_BENCH ")]
        if p.returncode != 0 or not lines:
            print(log[-2000:])
            raise SystemExit("a rank failed")
        out.append(json.loads(lines[-1][len("PEER_BENCH "):]))
    out.sort(key=lambda d: d["rank"])
    single = out[0].get("single_ms")
    worst = max(d["sharded_ms"] for d in out)
    same = all(np.abs(np.array(d["T"]) - np.array(out[0]["T"])).max() == 0 for d in out)
    eq1 = float(np.abs(np.array(out[0]["T"]) - np.array(out[0]["T_single"])).max()) if single else None
    print(json.dumps({"ranks": a.ranks, "devices": a.devices, "map": a.map, "scan": a.scan, "single_gpu_ms_per_registration": round(single, 4) if single else None,
                      "sharded_ms_per_registration": round(worst, 4), "speedup": round(single / worst, 3) if single else None, "launches_per_rank": [d["launches"] for d in out],
                      "aborts": [d["aborts"] for d in out], "aborts_single_phase": out[0].get("aborts_single_phase"), "evaluations": out[0]["evals"], "bit_identical_across_ranks": same, "max_abs_pose_difference_to_single_gpu": eq1,
                      "stages_us_rank0": out[0]["stages_us"]}))


if __name__ == "__main__":
    main()
