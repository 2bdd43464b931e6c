#!/usr/bin/env python
"""Concurrent aligns on one GPU: S engine handles, S host threads, align() only (clouds, covariances and the map are prepared once).
Prints aligns/s for S = 1, 2, 4, 8, the fraction of aligns that stayed on the one-launch route, watchdog aborts, placement aborts and the
grids the last aligns got -- once per environment given on the command line (ENV=VAL,ENV=VAL ...; each in its own process):
    python tools/r04_conc.py "" FVH_SHARE_BY_XCD=0 FVH_CONFINED_SLOT_PCT=50"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    if os.environ.get("CONC_IMPORT_TORCH") == "1":  # does a process that also holds torch's HIP context behave differently? (bench.py does)
        import torch
        torch.cuda.init(); torch.zeros(1, device="cuda")
    if os.environ.get("CONC_OMP_FIRST") == "1":     # ... or one whose main thread ran an OpenMP region of the oracle first (bench.py's cpu_baseline)?
        os.environ.setdefault("OMP_PROC_BIND", "close"); os.environ.setdefault("OMP_PLACES", "cores")
        from oracle import oracle as O
        from fast_gicp_amd import preprocess as pp
        t, s_ = pp.bundled_pair(os.path.join(ROOT, "data"))
        O.covariances_knn(t, 20, O.PLANE, threads=16)
        if os.environ.get("CONC_RESET_AFFINITY") == "1":
            os.sched_setaffinity(0, range(os.cpu_count()))
        print("  affinity of the main thread: %d CPUs" % len(os.sched_getaffinity(0)), flush=True)
    from fast_gicp_amd import capi, preprocess
    tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
    S_max, steps = 8, 60
    cores = []
    for _ in range(S_max):
        c = capi.VGICPCore(0)
        c.set_neighbor_search_method(capi.DIRECT27)
        c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
        c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
        c.align()
        cores.append(c)
    for S in (1, 2, 4, 8):
        launches = [[] for _ in range(S)]

        def loop(i):
            for _ in range(steps):
                launches[i].append(cores[i].align()["num_launches"])
        time.sleep(0.05)  # (the pool's concurrency estimate resets after 20 ms without overlap)
        th = [threading.Thread(target=loop, args=(i,)) for i in range(S)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        one = float(np.mean([np.mean(np.array(l) == 1) for l in launches]))
        print("  S=%d: %8.0f aligns/s  (%.1f us per align per stream)  one-launch %.2f  watchdog aborts %d  placement (wanted, aborts) %s  last grids %s" % (
            S, S * steps / el, el / steps * 1e6, one, sum(c.debug_persist_aborts() for c in cores), capi.debug_xcd_local(), [cores[i].debug_persist_grid()[0] for i in range(S)]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for spec in sys.argv[1:] or [""]:
            env = dict(os.environ)
            for kv in filter(None, spec.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            print("[%s]" % (spec or "default"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, timeout=300)
