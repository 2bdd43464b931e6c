#!/usr/bin/env python
"""Benchmark of the VGICP hot path on MI355X -- the reference's `100times_reuse` loop (src/align.cpp:87-101).

One "step" = one registration of the steady-state odometry pattern
    swapSourceAndTarget(); clearSource(); setInputTarget(same ptr -> no-op); setInputSource(next scan); align()
i.e. per step: 1 target voxel-map build (from the reused covariances) + covariance estimation of ONE cloud
(brute-force k-NN k=20 + PLANE) + one full LM solve.  Inputs are resident in HBM before the timed region.

N = 1 workload (BASELINE.json configs[1]): bundled 251370668/251371071 pair (HEAD preprocessing, 17,047 / 17,334 pts),
VGICP, DIRECT27, k_correspondences = 20, voxel resolution 1.0.
N > 1: every rank registers its own copy of the stream of scan pairs (registrations are independent units -> weak
scaling, no data-path collective); value = registrations of all ranks / max-over-ranks time.  The spatially sharded
single-registration path (RCCL all-reduce of the 28-value normal equations per evaluation) is measured separately and
reported under "sharded" when --gpus > 1.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="bundled17k", choices=["bundled17k", "synth100k", "synth1m", "lidar_stream"])
    ap.add_argument("--search", default=None, choices=["DIRECT1", "DIRECT7", "DIRECT27"])
    ap.add_argument("--cov", default="knn", choices=["knn", "rbf"])
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket the cost kernel with HIP events in the timed region")
    ap.add_argument("--sharded-deadline", type=int, default=240, help="--gpus > 1: seconds the extra spatially-sharded leg may take before it is abandoned")
    ap.add_argument("--streams", type=int, default=4, help="extra leg: S independent engine handles (own HIP streams, host threads) running the same loop concurrently on this GPU")
    ap.add_argument("--cpu-loops", type=int, default=0, help="oracle registrations to time (0 = auto-bound to ~15 s)")
    return ap.parse_args()


def make_workload(name):
    from fast_gicp_amd import preprocess
    if name == "bundled17k":
        tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
        return tgt, src, 1.0, "bundled 251370668<->251371071, ApproximateVoxelGrid 0.1 + origin filter (17,047/17,334 pts)"
    from tests import util
    if name == "synth100k":
        tgt, src, _ = util.synthetic_pair(100_000, 100_000, seed=42)
        return tgt, src, 0.5, "synthetic 100k<->100k LiDAR-like scene, seed 42"
    tgt, src, _ = util.synthetic_pair(1_000_000, 100_000, seed=44, extent=150.0)
    return tgt, src, 0.5, "synthetic 1M map <-> 100k scan, seed 44"


def run_with_deadline(fn, seconds):
    """(result, hung): fn() on a worker thread; hung=True if it is still running after `seconds` (the thread is abandoned)."""
    import threading
    box = {}
    th = threading.Thread(target=lambda: box.__setitem__("r", fn()), daemon=True)
    th.start()
    th.join(seconds)
    return box.get("r"), th.is_alive()


def finish(dist, hung):
    if hung:  # a worker thread sits in a collective that will never complete: leave without running any destructor
        sys.stdout.flush()
        os._exit(0)
    if dist is not None:
        dist.destroy_process_group()


def sharded_leg(args, dist, rank, world, local_rank, dev):
    """BASELINE.json configs[4]: 1M-point map <-> 100k-point scan, DIRECT7, res 0.5; the scan is sharded by spatial tile
    over the ranks, the 32-double normal-equation block is all-reduced by RCCL inside the device LM loop."""
    import torch
    from fast_gicp_amd import capi, distributed as D
    from tests import util
    try:
        torch.cuda.set_device(local_rank)  # this leg runs on a worker thread (run_with_deadline): the current device is per thread
        tgt, src, _ = util.synthetic_pair(1_000_000, 100_000, seed=44, extent=150.0)
        core = capi.VGICPCore(local_rank)
        core.set_resolution(0.5)
        core.set_neighbor_search_method(capi.DIRECT7)
        sh = D.ShardedVGICP(core, rank, world)
        ids = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        sh.init_device_collective(ids[0])
        t0 = time.perf_counter()
        sh.set_target(tgt)
        core.synchronize()
        map_ms = (time.perf_counter() - t0) * 1e3
        sh.set_source(src)
        r = sh.align()
        steps = 20
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = sh.align()
        dist.barrier(); torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        evals = r["num_linearize"] + r["num_error_evals"]
        core.comm_destroy()
        return {"workload": "synthetic 1M-point map <-> 100k-point scan, DIRECT7, res 0.5, source tiles over %d GPUs, replicated target map" % world,
                "aligns_per_sec": round(steps / float(el.item()), 3), "ms_per_align": round(float(el.item()) / steps * 1e3, 4), "evaluations_per_align": evals,
                "us_per_evaluation_incl_allreduce": round(float(el.item()) / steps / max(evals, 1) * 1e6, 2), "map_build_ms": round(map_ms, 2), "converged": bool(r["converged"]),
                "collective": "ncclAllReduce(32 x f64) on the engine stream, once per cost evaluation"}
    except Exception as e:  # the headline number must not depend on this leg
        return {"error": repr(e)}


def concurrent_leg(args, local_rank, d_clouds, n_pts, res, search, K):
    """Aggregate throughput of S independent registration streams on ONE GPU (one engine handle + HIP stream + host thread
    each). The headline `value` is the reference's sequential loop; this shows how much of the chip that loop leaves idle."""
    import threading
    from fast_gicp_amd import capi
    S, steps = args.streams, max(20, args.steps // 2)
    cores = []
    for _ in range(S):
        c = capi.VGICPCore(local_rank)
        c.set_resolution(res); c.set_neighbor_search_method(search); c.set_kernel_params(0.5, 2.5)
        c.set_precision(capi.COMPUTE_FP32 if args.precision == "fp32" else capi.COMPUTE_FP64)
        c.set_target_cloud_device(d_clouds[0].data_ptr(), n_pts[0], 3)
        c.find_target_neighbors(K); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
        c.set_source_cloud_device(d_clouds[1].data_ptr(), n_pts[1], 3)
        c.find_source_neighbors(K); c.calculate_source_covariances(capi.REG_PLANE)
        c.align()
        cores.append(c)

    def loop(c, n):
        nxt = 0
        for _ in range(n):
            c.swap_source_and_target()
            c.set_source_cloud_device(d_clouds[nxt].data_ptr(), n_pts[nxt], 3)
            c.find_source_neighbors(K); c.calculate_source_covariances(capi.REG_PLANE)
            c.align()
            nxt = 1 - nxt

    for c in cores:
        loop(c, 4)
    threads = [threading.Thread(target=loop, args=(c, steps)) for c in cores]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in cores:
        c.synchronize()
    el = time.perf_counter() - t0
    for c in cores:
        c.close()
    return {"streams": S, "steps_per_stream": steps, "registrations_per_sec": round(S * steps / el, 3), "note": "independent registrations, not the sequential reference loop"}


def stream_main(args):
    """BASELINE.json configs[3] (KITTI-style streaming, NDT D2D): the loop of src/kitti.cpp:95-128 with every stage on the
    device -- raw ~118k-point frame (resident in HBM) -> ApproximateVoxelGrid 0.25 (kitti.cpp:80-82) -> setInputSource ->
    align (NDTCuda defaults: D2D, DIRECT7, resolution 1.0) -> swapSourceAndTarget.  KITTI is not available offline: frames
    come from the 64-ring LiDAR simulator in tests/util.py (1 m ego-motion per frame); the sequence is walked back and
    forth so consecutive frames are always neighbours.  Single GPU only (a stream of dependent frames does not shard)."""
    import torch
    from fast_gicp_amd import capi
    from tests import util
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("lidar_stream is a single-GPU workload")
    F = 10
    kitti_dir = os.environ.get("FVH_KITTI_DIR")  # e.g. .../sequences/00/velodyne: the real frames instead of the simulator (kitti.cpp:22-69 format)
    if kitti_dir:
        frames = []
        for i in range(F):
            path = os.path.join(kitti_dir, "%06d.bin" % i)
            if not os.path.exists(path):
                break
            frames.append(np.ascontiguousarray(np.fromfile(path, np.float32)[: 1000000 // 4 * 4].reshape(-1, 4)[:, :3]))
        if len(frames) < 2:
            raise SystemExit("FVH_KITTI_DIR holds fewer than two %06d.bin frames")
        F = len(frames)
    else:
        frames = [util.lidar_frame(i) for i in range(F)]
    gpu = torch.device("cuda", 0)
    d_frames = [torch.from_numpy(f).to(gpu).contiguous() for f in frames]
    vg, ndt = capi.VoxelGrid(0), capi.NDTCore(0)
    ndt.set_distance_mode(capi.NDT_D2D); ndt.set_neighbor_search_method(capi.DIRECT7); ndt.set_resolution(1.0)
    ndt.set_precision(capi.COMPUTE_FP32 if args.precision == "fp32" else capi.COMPUTE_FP64)
    seq = list(range(F)) + list(range(F - 2, 0, -1))  # 0..F-1..1, repeated

    ptr, n = vg.filter_device(d_frames[0].data_ptr(), len(frames[0]), 0.25, vg.APPROXIMATE)
    ndt.set_target_cloud_device(ptr, n, 3)
    state = {"k": 0, "last": None, "n_ds": 0}

    def step():
        state["k"] += 1
        i = seq[state["k"] % len(seq)]
        ptr, n = vg.filter_device(d_frames[i].data_ptr(), len(frames[i]), 0.25, vg.APPROXIMATE)
        ndt.set_source_cloud_device(ptr, n, 3)
        state["last"] = ndt.align()
        ndt.swap_source_and_target()
        state["n_ds"] += n

    for _ in range(args.warmup):
        step()
    # accuracy on the first lap (not timed): per-frame relative pose vs the simulator's ground truth
    errs = []
    state["k"] = 0
    ptr, n = vg.filter_device(d_frames[0].data_ptr(), len(frames[0]), 0.25, vg.APPROXIMATE)
    ndt.set_target_cloud_device(ptr, n, 3)
    for i in range(1, F):
        step()
        if not kitti_dir:  # ground truth exists for the simulator only
            gt = np.linalg.inv(util.lidar_pose(i - 1)) @ util.lidar_pose(i)
            errs.append(util.pose_error(gt, state["last"]["T"])[0])
    state["k"] = F - 1
    profile = not args.no_profile
    ndt.profile_reset(); vg.profile_reset()
    ndt.profile_enable(False); vg.profile_enable(False)
    state["n_ds"] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_eval = 0
    for it in range(args.steps):
        if profile:
            ndt.profile_enable(it % 9 == 0); vg.profile_enable(it % 9 == 0)
        step()
        n_eval += state["last"]["num_linearize"] + state["last"]["num_error_evals"]
    ndt.synchronize(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ndt.profile_enable(False); vg.profile_enable(False)
    stage_ms, roofline = {}, None
    n_raw = int(np.mean([len(f) for f in frames]))
    n_ds = state["n_ds"] // max(args.steps, 1)
    if profile:
        for cls in ("cost", "voxelmap"):
            ms, n = ndt.profile_get(cls)
            if n:
                stage_ms[cls] = {"total_ms": round(ms, 3), "launches": n, "avg_us": round(ms / n * 1e3, 3)}
        ms, n = vg.profile_get()
        if n:
            stage_ms["downsample"] = {"total_ms": round(ms, 3), "launches": n, "avg_us": round(ms / n * 1e3, 3)}
            # dominant stage of this loop: the filter. Algorithmic bytes: read N x 12 B, write M x 12 B
            b = n_raw * 12 + n_ds * 12
            ach = b / (ms / n * 1e-3) / 1e9
            roofline = {"kernel": "voxel-grid filter (9 launches: keys, 1 radix pass, mark, scan, emit)", "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(ach / 8000.0, 5), "traffic": None, "algorithmic_bytes_per_launch": b, "avg_launch_us": round(ms / n * 1e3, 3), "launches": n,
                        "note": "1.4 MB of input per frame: launch/latency bound (a chain of 9 dependent small kernels + one D2H count), not HBM bound"}
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = min(os.cpu_count() or 1, 64)
        g = O.NDT(threads=cores, mode=O.D2D, search=O.DIRECT7)
        loops = args.cpu_loops or 40
        g.set_target(O.approx_voxelgrid(frames[0], 0.25))
        t1 = time.perf_counter()
        for k in range(1, loops + 1):
            g.set_source(O.approx_voxelgrid(frames[seq[k % len(seq)]], 0.25))
            g.align()
            g.swap()
        cpu = {"value": round(loops / (time.perf_counter() - t1), 3), "unit": "registrations/sec", "cores": cores, "kind": "port",
               "sample": "%d frames of the same loop: oracle ApproximateVoxelGrid (1 thread, as PCL) + oracle NDT D2D (OpenMP, %d threads)" % (loops, cores)}
    out = {"metric": "registrations/sec (frame-by-frame odometry, kitti.cpp loop incl. downsampling)", "value": round(args.steps / elapsed, 3), "unit": "registrations/sec", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64" if args.precision == "fp64" else "f32", "data": "KITTI" if kitti_dir else "synthetic",
           "config": {"workload": "%s, %d raw pts/frame -> ApproximateVoxelGrid 0.25 -> %d pts; NDT D2D, DIRECT7, res 1.0; %d-frame sequence walked back and forth" % ("KITTI frames from FVH_KITTI_DIR" if kitti_dir else "simulated 64-ring LiDAR", n_raw, n_ds, F),
                      "method": "NDT_D2D", "neighbor_search": "DIRECT7", "voxel_resolution": 1.0, "parallelism": "single GPU"},
           "per_registration": {"cost_evaluations": n_eval / args.steps, "converged": bool(state["last"]["converged"])},
           "accuracy": ({"max_frame_translation_error_m": round(float(max(errs)), 4), "mean_frame_translation_error_m": round(float(np.mean(errs)), 4)} if errs else None),
           "roofline": roofline, "cpu_baseline": cpu, "stages": stage_ms}
    print(json.dumps(out))


def main():
    args = parse()
    if args.workload == "lidar_stream":
        return stream_main(args)
    import torch
    from fast_gicp_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # test-only knobs to walk the N > 1 control flow on a single-GPU box: all ranks on device 0, gloo for the barrier
    share_gpu = os.environ.get("FVH_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("FVH_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
        os.environ.setdefault("FVH_PERSISTENT", "0")  # several processes on one GPU: their persistent grids would starve each other (the watchdog recovers, slowly)
        os.environ.setdefault("FVH_SORT_MODE", "1")   # same for the cooperative sort
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    tgt, src, res, desc = make_workload(args.workload)
    if args.search is None:
        args.search = "DIRECT7" if args.workload == "synth1m" else "DIRECT27"
    search = {"DIRECT27": capi.DIRECT27, "DIRECT7": capi.DIRECT7, "DIRECT1": capi.DIRECT1}[args.search]
    K = 20
    gpu = torch.device("cuda", local_rank)
    d_clouds = [torch.from_numpy(tgt).to(gpu).contiguous(), torch.from_numpy(src).to(gpu).contiguous()]  # inputs resident in HBM
    n_pts = [len(tgt), len(src)]

    core = capi.VGICPCore(local_rank)
    core.set_resolution(res)
    core.set_neighbor_search_method(search)
    core.set_kernel_params(0.5, 2.5)  # align.cpp:210 setKernelWidth(0.5) -> max_dist 2.5
    core.set_precision(capi.COMPUTE_FP32 if args.precision == "fp32" else capi.COMPUTE_FP64)

    def estimate_cov(which):
        if args.cov == "knn":
            getattr(core, "find_%s_neighbors" % which)(K)
            getattr(core, "calculate_%s_covariances" % which)(capi.REG_PLANE)
        else:
            getattr(core, "calculate_%s_covariances_rbf" % which)(capi.REG_PLANE)

    # "single" (align.cpp:58-68): both clouds from scratch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    core.set_target_cloud_device(d_clouds[0].data_ptr(), n_pts[0], 3)
    estimate_cov("target")
    core.create_target_voxelmap()
    core.set_source_cloud_device(d_clouds[1].data_ptr(), n_pts[1], 3)
    estimate_cov("source")
    first = core.align()
    single_ms = (time.perf_counter() - t0) * 1e3
    fitness = core.fitness_score(first["T"].astype(np.float32).astype(np.float64))

    state = {"next": 0, "last": first}  # after "single": target = cloud 0, source = cloud 1

    if args.workload == "synth1m":
        # map-vs-scan localisation (BASELINE configs[4] shape): the 1M-point map stays the target, every step registers a scan
        def step():
            core.set_source_cloud_device(d_clouds[1].data_ptr(), n_pts[1], 3)
            estimate_cov("source")
            state["last"] = core.align()
            state["next"] = 0
    else:
        def step():
            # reg.swapSourceAndTarget(); reg.clearSource(); reg.setInputTarget(target_) [no-op]; reg.setInputSource(source_); reg.align()
            core.swap_source_and_target()
            i = state["next"]
            core.set_source_cloud_device(d_clouds[i].data_ptr(), n_pts[i], 3)
            estimate_cov("source")
            state["last"] = core.align()
            state["next"] = 1 - i

    for _ in range(args.warmup):
        step()

    profile = not args.no_profile
    # HIP-event bracketing costs ~20 % on a registration it is applied to: sample every 9th registration of the timed region
    # (odd on purpose: the loop alternates between the two directions of the pair, which need 6 and 8 LM transitions)
    PROFILE_EVERY = 9
    core.profile_reset()
    core.profile_enable(False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    n_lin = n_err = n_launch = 0
    for it in range(args.steps):
        if profile:
            core.profile_enable(it % PROFILE_EVERY == 0)
        step()
        n_lin += state["last"]["num_linearize"]
        n_err += state["last"]["num_error_evals"]
        n_launch += state["last"]["num_launches"]
    barrier()
    elapsed = time.perf_counter() - t0
    core.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sharded, sharded_hung = None, False
    if world > 1 and not share_gpu:
        # the headline number is already measured: a stuck collective in this extra leg must not take the JSON line with it
        sharded, sharded_hung = run_with_deadline(lambda: sharded_leg(args, dist, rank, world, local_rank, dev), args.sharded_deadline)
        if sharded_hung:
            sharded = {"error": "sharded leg did not finish within %d s on rank %d" % (args.sharded_deadline, rank)}
    if rank != 0:
        finish(dist, sharded_hung)
        return

    total_regs = args.steps * world
    value = total_regs / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel class of the LM loop (cost evaluation) ----
    n_off = {"DIRECT27": 27, "DIRECT7": 7, "DIRECT1": 1}[args.search]
    n_c = core.get_num_correspondences()  # valid (source, voxel) pairs of the last linearisation
    n_src = n_pts[1 - state["next"]]
    # B_eval = N_s*48 + N_s*N_off*16 + N_c*52 (+172 B out)
    bytes_eval = n_src * 48 + n_src * n_off * 16 + n_c * 52 + 172
    roofline = None
    stage_ms = {}
    if profile:
        cost_ms, cost_n = core.profile_get("cost")
        for cls in ("cost", "knn", "cov", "rbf", "voxelmap"):
            ms, n = core.profile_get(cls)
            if n:
                stage_ms[cls] = {"total_ms": round(ms, 3), "launches": n, "avg_us": round(ms / n * 1e3, 3)}
        if cost_n:
            avg_s = cost_ms / cost_n * 1e-3
            # a registration's cost evaluations move (n_lin + n_err) * B_eval algorithmic bytes. One launch of the
            # persistent LM kernel runs ALL of them (kernel_launches_lm == 1); the multi-launch path spreads them over its launches.
            evals_per_launch = (n_lin + n_err) / max(n_launch, 1)
            bytes_launch = bytes_eval * evals_per_launch
            achieved = bytes_launch / avg_s / 1e9
            traffic = None
            try:  # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs; FETCH doubled per the gfx950 note)
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_cost_kernel.json")))
                traffic = pmc.get(args.workload + ("_persistent" if n_launch == args.steps else ""), {}).get("hbm_bytes_per_launch")
            except Exception:
                pass
            kname = "cost_kernel<%s,VGICP,%s>" % ("double" if args.precision == "fp64" else "float", "persistent" if n_launch == args.steps else "per-transition")
            roofline = {"kernel": kname, "bound": "hbm", "achieved": round(achieved, 2),
                        "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic, "algorithmic_bytes_per_launch": int(bytes_launch),
                        "algorithmic_bytes_per_evaluation": bytes_eval, "evaluations_per_launch": round(evals_per_launch, 3),
                        "avg_launch_us": round(avg_s * 1e6, 3), "launches": cost_n,
                        "note": ("17k-point working set (~3 MB) is L2/Infinity-Cache resident and the fused trips share their loads between the trial evaluation and the next "
                                 "linearisation: this kernel is bound by the latency chain of its %d barrier-separated trips, not by HBM; `traffic` (PMC) is what actually reaches HBM"
                                 % round((n_err / args.steps) + 1) if args.workload == "bundled17k"
                                 else "launch average over the LM launches of the timed region")}
    if args.cov == "knn" and "knn" in stage_ms:
        n = n_src
        flops = 8.0 * n * n
        # the k-NN is culled (it evaluates ~10 tiles x 64 pairs per query, not N pairs): this is the rate a full
        # 8*N^2-flop brute-force sweep would need to match it, not a VALU utilisation figure
        stage_ms["knn"]["bruteforce_equivalent_tflops"] = round(flops / (stage_ms["knn"]["avg_us"] * 1e-6) / 1e12, 3)

    # ---- CPU baseline: the oracle (fp64 OpenMP restatement of FastVGICP) on this box's host cores ----
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as O
        cores = min(os.cpu_count() or 1, 64)
        g = O.FastVGICP(threads=cores, search={"DIRECT27": O.DIRECT27, "DIRECT7": O.DIRECT7, "DIRECT1": O.DIRECT1}[args.search], resolution=res,
                        cov_mode=1 if args.cov == "rbf" else 0, kernel_width=0.5, kernel_max_dist=2.5)
        g.bench(tgt, src, 0, 1)  # "single": primes both clouds
        t1 = time.perf_counter()
        g.bench(tgt, src, 2, 2)
        per = (time.perf_counter() - t1) / 2
        loops = args.cpu_loops or int(max(4, min(100, 15.0 / max(per, 1e-3))))
        ms, _ = g.bench(tgt, src, 2, loops)
        cpu = {"value": round(loops / (ms * 1e-3), 3), "unit": "registrations/sec", "cores": cores, "kind": "port",
               "sample": "%d iterations of the 100times_reuse loop on the same pair/config (oracle/liboracle.so, OpenMP, %d threads)" % (loops, cores)}
        if args.workload == "bundled17k":  # OMP_NUM_THREADS in {1, nproc}
            g1 = O.FastVGICP(threads=1, search={"DIRECT27": O.DIRECT27, "DIRECT7": O.DIRECT7, "DIRECT1": O.DIRECT1}[args.search], resolution=res,
                             cov_mode=1 if args.cov == "rbf" else 0, kernel_width=0.5, kernel_max_dist=2.5)
            g1.bench(tgt, src, 0, 1)
            ms1, _ = g1.bench(tgt, src, 2, 5)
            cpu["single_thread"] = {"value": round(5 / (ms1 * 1e-3), 3), "unit": "registrations/sec", "cores": 1, "sample": "5 iterations of the same loop, 1 thread"}

    conc = None
    if args.streams > 1 and world == 1 and args.cov == "knn" and args.workload != "synth1m":
        try:
            conc = concurrent_leg(args, local_rank, d_clouds, n_pts, res, search, K)
        except Exception as ex:
            conc = {"error": repr(ex)}
    out = {
        "metric": "registrations/sec (100-iter reuse) + final fitness_score; achieved HBM GB/s",
        "value": round(value, 3), "unit": "registrations/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == "fp64" else "f32", "data": "bundled scans (real LiDAR)" if args.workload == "bundled17k" else "synthetic",
        "config": {"workload": desc, "method": "VGICP", "neighbor_search": args.search, "k_correspondences": K, "covariance": args.cov, "regularization": "PLANE",
                   "voxel_resolution": res, "parallelism": "1 registration stream per GPU" if world > 1 else "single GPU"},
        "fitness_score": round(fitness, 6), "single_ms": round(single_ms, 3),
        "per_registration": {"linearize": n_lin / args.steps, "error_evals": n_err / args.steps, "kernel_launches_lm": n_launch / args.steps, "converged": bool(state["last"]["converged"]),
                             "persistent_launches_aborted_by_watchdog": core.debug_persist_aborts()},
        "roofline": roofline, "cpu_baseline": cpu, "stages": stage_ms, "profiled_timed_region": ("every %dth registration" % PROFILE_EVERY) if profile else False,
    }
    if sharded is not None:
        out["sharded"] = sharded
    if conc is not None:
        out["concurrent_streams"] = conc
    print(json.dumps(out), flush=True)
    finish(dist, sharded_hung)


if __name__ == "__main__":
    main()
