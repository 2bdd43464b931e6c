"""How do S registration handles of ONE process share the GPU? `python -m fast_gicp_amd.concurrency_probe [S ...]` prepares S engine
handles on the bundled pair (clouds, covariances, target map), then runs align() alone on S host threads and prints one JSON line:
aligns/s per S, the fraction of aligns that stayed on the one-launch (persistent) route, the grids the SlotPool granted.

It runs in a process of its own on purpose (bench.py calls it as a subprocess): a process that also holds torch's HIP context measures
25-30 % lower at four threads (12.3k instead of 16.8k aligns/s on the MI355X of round 4, tools/r04_conc.py), and threads started from a
thread that has run an OpenMP region under OMP_PROC_BIND inherit its one-place CPU mask."""
import json
import os
import sys
import threading
import time

import numpy as np

from . import capi, preprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(stream_counts=(1, 2, 4, 8), steps=60, device=0):
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except (AttributeError, OSError):
        pass
    tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
    cores = []
    for _ in range(max(stream_counts)):
        c = capi.VGICPCore(device)
        c.set_neighbor_search_method(capi.DIRECT27)
        c.set_target_cloud(tgt); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
        c.set_source_cloud(src); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
        c.align()
        cores.append(c)
    out = {"workload": "bundled 17k pair, DIRECT27, align() only on prepared handles", "steps_per_stream": steps, "streams": {}}
    for S in stream_counts:
        launches = [[] for _ in range(S)]

        def loop(i):
            for _ in range(steps):
                launches[i].append(cores[i].align()["num_launches"])
        time.sleep(0.05)  # (the pool's concurrency estimate starts afresh after 20 ms without overlap)
        th = [threading.Thread(target=loop, args=(i,)) for i in range(S)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        out["streams"][str(S)] = {"aligns_per_sec": round(S * steps / el, 1), "us_per_align_per_stream": round(el / steps * 1e6, 1),
                                  "one_launch_fraction": round(float(np.mean([np.mean(np.array(l) == 1) for l in launches])), 3),
                                  "grids": [cores[i].debug_persist_grid()[0] for i in range(S)]}
    out["persistent_launches_aborted_by_watchdog"] = sum(c.debug_persist_aborts() for c in cores)
    out["xcd_local_wanted_and_placement_aborts"] = list(capi.debug_xcd_local())
    one = out["streams"].get("1", {}).get("aligns_per_sec")
    if one:
        for k, v in out["streams"].items():
            v["ratio_to_one_handle"] = round(v["aligns_per_sec"] / one, 3)
    for c in cores:
        c.close()
    return out


if __name__ == "__main__":
    counts = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 4, 8)
    print(json.dumps(run(counts)), flush=True)
