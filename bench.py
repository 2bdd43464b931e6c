#!/usr/bin/env python
"""Benchmark of the VGICP / NDT hot path on MI355X -- the reference's `100times_reuse` loop (src/align.cpp:87-101).

One "step" = one registration of the steady-state odometry pattern
    swapSourceAndTarget(); clearSource(); setInputTarget(same ptr -> no-op); setInputSource(next scan); align()
i.e. per step: 1 target voxel-map build (from the reused covariances) + covariance estimation of ONE cloud
(brute-force k-NN k=20 + PLANE) + one full LM solve.

`value` (the contract's number): inputs resident in HBM before the timed region (fvh_vgicp_set_source_cloud_device).
`host_clouds_in`: the SAME loop fed the way src/align.cpp:94-96 feeds it -- a host cloud per registration through
fvh_vgicp_set_source_cloud (one H2D inside the timed region): the PCIe-inclusive, reference-exact rate, measured in the same run.

N = 1 headline workload (BASELINE.json configs[1]): bundled 251370668/251371071 pair (HEAD preprocessing, 17,047 / 17,334 pts),
VGICP, DIRECT27, k_correspondences = 20, voxel resolution 1.0.  The default invocation also measures, time-boxed, the other
single-GPU configurations of BASELINE.json and reports them under "configs", each with its own roofline and cpu_baseline:
    synth100k_rbf  configs[2]  synthetic 100k <-> 100k, res 0.5, RBF covariances 0.5/2.5, DIRECT27
    synth1m        configs[4]  synthetic 1M-point map <-> 100k-point scan, res 0.5, DIRECT7 (1-GPU form of the sharded config)
    lidar_stream   configs[3]  frame-by-frame NDT D2D on ~118k-point simulated LiDAR frames incl. on-device downsampling
N > 1: every rank registers its own copy of the stream of scan pairs (registrations are independent units -> weak
scaling, no data-path collective); value = registrations of all ranks / max-over-ranks time.  The spatially sharded
single-registration path (all-reduce of the normal equations per evaluation) is measured separately and reported under
"sharded" together with the 1-GPU time of the same registration.

Output (rank 0): the LAST stdout line is ONE compact JSON object (< 4 KB: the contract's keys, `roofline`, `cpu_baseline`, one short
entry per extra configuration -- compact_line()). The full detail (stages, thread sweeps, notes, per-route sharded detail) goes to
bench_detail.json beside this script and to stderr, never to stdout.
The timed region is run REPEATS times (each: exactly --steps steps between barrier + synchronize); `ms_per_step` / `value` are the median
repeat, min / max are reported beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_START = time.perf_counter()
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc.json")
# the cpu_baseline legs (oracle, OpenMP): threads next to each other -- they share the voxel map and the kd-tree through the caches
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="bundled17k", choices=["bundled17k", "synth100k", "synth1m", "lidar_stream", "fgicp17k"])
    ap.add_argument("--search", default=None, choices=["DIRECT1", "DIRECT7", "DIRECT27"])
    ap.add_argument("--cov", default="knn", choices=["knn", "rbf", "kdtree"])
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"])
    ap.add_argument("--configs", default=None, help="comma list of extra configurations measured after the headline (default: all three when the headline is the default "
                    "bundled17k run on one GPU; 'none' to skip): synth100k_rbf,synth1m,lidar_stream")
    ap.add_argument("--time-box", type=float, default=170.0, help="seconds after which no further extra configuration is started")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-clouds-in (PCIe-inclusive) repetition of the timed loop")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket the cost kernel with HIP events in the timed region")
    ap.add_argument("--no-pipelined-leg", action="store_true", help="skip the pipelined side legs (bundled17k / synth100k: fvh_vgicp_prepare_source_device beside the LM kernel; lidar_stream: the two-stage frame pipeline; their LM launches overlap with the next frame's preparation and are slower: "
                    "a rocprofv3 --stats average over the whole command mixes the two populations, see profiles/README.md)")
    ap.add_argument("--sharded-deadline", type=int, default=240, help="--gpus > 1: seconds the extra spatially-sharded leg may take before it is abandoned")
    ap.add_argument("--streams", type=int, default=4, help="extra leg: S independent engine handles (own HIP streams, host threads) running the same loop concurrently on this GPU")
    ap.add_argument("--cpu-loops", type=int, default=0, help="oracle registrations to time (0 = auto-bound)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the headline's cpu_baseline leg (thread sweep included)")
    return ap.parse_args()


def make_workload(name):
    from fast_gicp_amd import preprocess, workloads
    if name == "bundled17k":
        tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
        return tgt, src, 1.0, "bundled 251370668<->251371071, ApproximateVoxelGrid 0.1 + origin filter (17,047/17,334 pts)"
    if name == "synth100k":
        tgt, src, _ = workloads.synthetic_pair(100_000, 100_000, seed=42)
        return tgt, src, 0.5, "synthetic 100k<->100k LiDAR-like scene, seed 42"
    tgt, src, _ = workloads.synthetic_pair(1_000_000, 100_000, seed=44, extent=150.0)
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


def csrc_sha():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fast_gicp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


REPEATS = 5
COMPACT_LIMIT = 4096
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")

_ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us", "flop_frac")


def median_of(times):
    """(median, min, max) of the repeat times; the median of an even count is the mean of the middle two."""
    t = sorted(times)
    n = len(t)
    med = t[n // 2] if n % 2 else 0.5 * (t[n // 2 - 1] + t[n // 2])
    return med, t[0], t[-1]


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _compact_roofline(r):
    if not isinstance(r, dict):
        return None
    if "error" in r:
        return {"error": _short(r["error"], 120)}
    out = {k: r[k] for k in _ROOFLINE_KEYS if k in r}
    if isinstance(out.get("unit"), str):
        out["unit"] = out["unit"].split(" (")[0]
    out["kernel"] = _short(out.get("kernel"), 80)
    return out


def _compact_cpu(c):
    if not isinstance(c, dict):
        return None
    out = {k: c[k] for k in ("value", "unit", "cores", "kind") if k in c}
    out["sample"] = _short(c.get("sample"), 150)
    if "host_cores" in c:
        out["host_cores"] = c["host_cores"]
    return out


def _compact_config(c):
    """one extra configuration -> {value, ms_per_step, frac, avg_launch_us} (+ the pipelined rate of the LiDAR loop, + cores of a CPU-only entry)"""
    if not isinstance(c, dict):
        return None
    if "error" in c or "skipped" in c:
        return {k: _short(c[k], 100) for k in ("error", "skipped") if k in c}
    out = {}
    for k in ("value", "unit", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "steps", "cores", "kind", "fitness_score", "host_kdtree_opt_in", "gicp_reuse", "fitness_us"):
        if k in c:
            out[k] = c[k]
    r = c.get("roofline")
    if isinstance(r, dict) and "frac" in r:
        out["frac"] = r["frac"]
        out["avg_launch_us"] = r.get("avg_launch_us")
        out["traffic"] = r.get("traffic")
    cb = c.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        out["cpu_baseline"] = {"value": cb["value"], "cores": cb.get("cores"), "kind": cb.get("kind")}
    p = c.get("pipelined")
    if isinstance(p, dict):
        out["pipelined"] = p.get("registrations_per_sec", p.get("value", _short(p.get("error"), 80)))
    return out


def compact_line(detail):
    """The line the driver parses: the contract's keys + roofline + cpu_baseline + one short entry per extra configuration, as ONE strict-JSON
    line shorter than COMPACT_LIMIT bytes whatever the detail dictionary holds (optional parts are dropped, longest first, until it fits)."""
    d = detail
    cfg = d.get("config") or {}
    out = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["metric"] = _short(out["metric"], 140)
    # (single_ms / ms_per_reg_100times: the other two rows align.cpp prints per method -- "single" :58-68 and "100times" :73-83, per registration)
    for k in ("repeats", "ms_per_step_min", "ms_per_step_max", "fitness_score", "single_ms", "ms_per_reg_100times", "speedup_vs_one_gpu"):
        if k in d:
            out[k] = d[k]
    out["config"] = {k: _short(v, 160) for k, v in cfg.items() if k in ("workload", "method", "neighbor_search", "k_correspondences", "covariance", "regularization", "voxel_resolution",
                                                                         "parallelism", "loop", "inputs", "route")}
    out["roofline"] = _compact_roofline(d.get("roofline"))
    out["cpu_baseline"] = _compact_cpu(d.get("cpu_baseline"))
    optional = []
    if isinstance(d.get("host_clouds_in"), dict):
        out["host_clouds_in"] = {k: d["host_clouds_in"].get(k) for k in ("value", "ms_per_step")}
        optional.append("host_clouds_in")
    if isinstance(d.get("pipelined"), dict) and "value" in d["pipelined"]:
        out["pipelined"] = {k: d["pipelined"].get(k) for k in ("value", "ms_per_step")}
        optional.append("pipelined")
    if isinstance(d.get("n1_same_workload"), dict):
        out["n1_same_workload"] = {k: d["n1_same_workload"].get(k) for k in ("value", "ms_per_step", "steps")}
    if isinstance(d.get("replicas_17k"), dict):
        out["replicas_17k"] = _compact_config(d["replicas_17k"])
        optional.append("replicas_17k")
    if isinstance(d.get("per_registration"), dict):
        out["per_registration"] = {k: v for k, v in d["per_registration"].items() if not isinstance(v, (dict, list, str))}
        optional.append("per_registration")
    if isinstance(d.get("concurrent_streams"), dict) and "registrations_per_sec" in d["concurrent_streams"]:
        out["concurrent_streams"] = {"streams": d["concurrent_streams"].get("streams"), "value": d["concurrent_streams"]["registrations_per_sec"]}
        optional.append("concurrent_streams")
    if isinstance(d.get("configs"), dict):
        out["configs"] = {k: _compact_config(v) for k, v in d["configs"].items()}
    if "sharded_headline_error" in d:
        out["sharded_headline_error"] = _short(d["sharded_headline_error"], 160)
    for k in ("bench_wall_s", "detail"):
        if k in d:
            out[k] = d[k]

    def dumps(o):
        return json.dumps(o, separators=(",", ":"), allow_nan=False)

    def clean(o):  # strict JSON: NaN / Infinity -> null
        if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
            return None
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (np.floating, np.integer, np.bool_)):
            return clean(o.item())
        return o

    out = clean(out)
    line = dumps(out)
    for k in optional[::-1] + ["configs"]:  # (never reached with the shapes this script builds; the bound must hold for ANY detail)
        if len(line) < COMPACT_LIMIT:
            break
        if k == "configs" and isinstance(out.get("configs"), dict):
            out["configs"] = {n: ({"value": c.get("value")} if isinstance(c, dict) else None) for n, c in out["configs"].items()}
        else:
            out.pop(k, None)
        line = dumps(out)
    if len(line) >= COMPACT_LIMIT:
        out.pop("configs", None)
        out["config"] = {"workload": _short((out.get("config") or {}).get("workload"), 100)}
        line = dumps(out)
    assert len(line) < COMPACT_LIMIT and "\n" not in line
    return line


def emit(detail):
    """bench_detail.json + stderr get everything; stdout gets the compact line, last."""
    detail["detail"] = "bench_detail.json (beside bench.py) + stderr"
    try:
        with open(DETAIL_FILE, "w") as f:
            json.dump(detail, f)
    except OSError as ex:
        detail["detail"] = "stderr only (%r)" % ex
    sys.stdout.flush()
    print("bench_detail: " + json.dumps(detail), file=sys.stderr, flush=True)
    print(compact_line(detail), flush=True)


_PMC = {}


def pmc_entry(key):
    """(entry, source) of the committed PMC passes of this round (profiles/r06_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ group
    in separate runs of THIS command, tools/r03_artifacts.sh + tools/pmc_collect.py; FETCH doubled per the gfx950 note of
    MI355X_MICROARCH.md). PMC counters cannot be collected inside a timed run, so the numbers are a committed measurement -- stamped
    with the commit and a hash of fast_gicp_amd/csrc/ they were taken at, and NOT quoted once the kernels have changed."""
    if not _PMC:
        try:
            _PMC.update(json.load(open(PMC_FILE)))
        except Exception:
            _PMC["_meta"] = {}
        _PMC["_stale"] = _PMC.get("_meta", {}).get("csrc_sha") != csrc_sha()
    meta = _PMC.get("_meta", {})
    if not meta:
        return None, "no PMC pass committed for this round (profiles/r06_pmc.json)"
    if _PMC["_stale"]:
        return None, "PMC pass of commit %s is older than fast_gicp_amd/csrc/ (hash %s != %s): not quoted" % (meta.get("commit"), meta.get("csrc_sha"), csrc_sha())
    return _PMC.get(key), "committed PMC pass of commit %s (profiles/r06_pmc.json, csrc hash %s == this tree)" % (meta.get("commit"), meta.get("csrc_sha"))


def pmc_traffic(key):
    e, src = pmc_entry(key)
    return (e or {}).get("hbm_bytes_per_launch"), src


VALU_PEAK_LANE_OPS = 78.6e12   # 1024 SIMDs x 64 lanes / 2 cycles x 2.4 GHz: one wave64 VALU instruction per SIMD every 2 cycles (MI355X_MICROARCH.md)
VALU_PEAK_FLOPS = 157.3e12     # the same, 2 flop per lane-op (fma): the guide's fp32 vector peak
VALU_PROBE_LANE_OPS = 50.9e12  # what a pure v_fma_f32 stream of 8 waves per SIMD sustained on this chip (tools/probes/probe_valu_rate.hip, profiles/r05_valu_rate.txt)


def valu_roofline(key, kernel, avg_us, n_queries, n_candidates):
    """the O(N^2)-class kernels (exact k-NN, RBF sweep) are fp32-VALU bound, not HBM bound. `frac` = the share of the peak VALU
    ISSUE rate (SQ_INSTS_VALU x 2 cycles / SIMD cycles of the launch, from the committed SQ pass: <= 1 by construction); `flop_frac` = the
    useful arithmetic -- candidate distances and box tests the culled search actually evaluated (tools/pair_counts.py) x flop each / launch
    time / 157.3 TFLOP/s: most issued instructions of these kernels are selection (64-bit key compares, DPP moves, ballots), not flops."""
    e, src = pmc_entry(key)
    sq = (e or {}).get("sq") or {}
    pairs = (e or {}).get("pairs") or {}
    insts, t_us = (sq.get("counters_per_launch") or {}).get("SQ_INSTS_VALU"), sq.get("avg_launch_us_in_this_pass")
    rate = None if not insts or not t_us else insts * 64 / (t_us * 1e-6)
    flop = pairs.get("flop_per_launch")
    flop_rate = None if not flop or not avg_us else flop / (avg_us * 1e-6)
    return {"kernel": kernel, "bound": "valu", "achieved": None if rate is None else round(rate / 1e12, 3), "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s",
            "frac": sq.get("valu_issue_utilisation"),
            "frac_definition": "VALU issue utilisation: SQ_INSTS_VALU x 2 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) of the PMC pass -- the share of the peak rate of one wave64 "
                               "VALU instruction per SIMD every 2 cycles; <= 1 by construction (DPP, 64-bit and transcendental instructions take longer than 2 cycles)",
            "frac_of_measured_fma_issue_rate": None if rate is None else round(rate / VALU_PROBE_LANE_OPS, 4),
            "flop_frac": None if flop_rate is None else round(flop_rate / VALU_PEAK_FLOPS, 5),
            "flop_per_launch": flop, "achieved_tflops": None if flop_rate is None else round(flop_rate / 1e12, 3), "peak_tflops": VALU_PEAK_FLOPS / 1e12,
            "candidates_per_query": pairs.get("candidates_per_query"), "box_tests_per_launch": pairs.get("box_tests_per_launch"),
            "valu_busy_frac_of_wave_cycles": sq.get("valu_busy_frac_of_wave_cycles"), "waiting_frac_of_wave_cycles": sq.get("waiting_frac_of_wave_cycles"),
            "valu_insts_per_query": sq.get("valu_insts_per_wave"), "counters_source": src,
            "pair_evaluations_per_sec_full_sweep_equivalent": round(float(n_queries) * n_candidates / (avg_us * 1e-6), 1),
            "note": "one query per wave, culled by two levels of tile boxes: the full-sweep-equivalent rate is what a brute-force N x N kernel would have to sustain, "
                    "not arithmetic this kernel performs; flop_frac counts the arithmetic it does perform"}


def thread_counts():
    n = os.cpu_count() or 1
    return sorted(set(t for t in (1, 8, 16, 32, 64, n) if 1 <= t <= n))


def cpu_baseline_vgicp(tgt, src, res, search, cov, budget_s, mode="reuse", counts=None):
    """The oracle (fp64 OpenMP restatement of FastVGICP) on this box's host cores: the loop of the workload, thread counts
    {1, 8, 16, 32, nproc} swept on a short sample, the best one timed on the rest of the budget and reported."""
    from oracle import oracle as O
    osearch = {"DIRECT27": O.DIRECT27, "DIRECT7": O.DIRECT7, "DIRECT1": O.DIRECT1}[search]

    def make(threads):
        g = O.FastVGICP(threads=threads, search=osearch, resolution=res, cov_mode=1 if cov == "rbf" else 0, kernel_width=0.5, kernel_max_dist=2.5)
        return g

    def time_loops(g, loops):
        if mode == "reuse":  # align.cpp:87-101
            ms, _ = g.bench(tgt, src, 2, loops)
            return ms * 1e-3
        t0 = time.perf_counter()  # scan-to-map localisation: the map stays the target, a fresh scan is registered every step
        for _ in range(loops):
            g.clear_source(); g.set_source(src); g.align()
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    sweep = {}
    swept = []
    best = None
    counts = counts or thread_counts()
    for th in counts:  # ascending: the sweep stops once more threads stop paying (a 256-thread OpenMP team on 17k-item loops is 40x slower than 64)
        g = make(th)
        if mode == "reuse":
            g.bench(tgt, src, 0, 1)  # "single": primes both clouds (covariances, kd-trees)
        else:
            g.set_target(tgt); g.set_source(src); g.align()
        t = time_loops(g, 1)
        loops = int(max(1, min(10, (budget_s / (3 * len(counts))) / max(t, 1e-3))))
        t = min(t, time_loops(g, loops) / loops)
        sweep[th] = round(1.0 / t, 3)
        swept.append((t, th, g))
        if best is None or t < best[1]:
            best = (th, t, g)
        elif t > 1.5 * best[1]:
            break
        if time.perf_counter() - t_begin > budget_s * 0.75:
            break
    # the three best thread counts of the (short) sweep are all timed on the rest of the budget; the fastest one is reported. The
    # sweep's one-to-ten-iteration samples over-read the wide teams on the GPU boxes' hosts (64 threads: 62 registrations/s in the sweep,
    # 13 over ten iterations; 16 threads: 41 both times): a sweep sample alone must not decide which configuration stands for the host
    swept.sort(key=lambda x: x[0])
    left = budget_s - (time.perf_counter() - t_begin)
    finals = []
    for t, th, g in swept[:3]:
        loops = int(max(10, min(100, (left / len(swept[:3])) / max(t, 1e-3))))  # (never fewer than 10 iterations, whatever the budget says)
        el = time_loops(g, loops)
        finals.append((loops / el, th, loops))
    rate, th, loops = max(finals)
    return {"value": round(rate, 3), "unit": "registrations/sec", "cores": th, "kind": "port",
            "sample": "%d iterations of the %s on the same pair/config (oracle/liboracle.so, OpenMP, %d threads = the fastest of the sweep's three best thread counts, all re-timed)" % (
                loops, "100times_reuse loop" if mode == "reuse" else "scan-to-map loop (map prepared once)", th),
            "confirmed_registrations_per_sec": {str(c): round(r, 3) for r, c, _ in finals},
            "thread_sweep_registrations_per_sec": sweep, "host_cores": os.cpu_count(), "omp_proc_bind": os.environ.get("OMP_PROC_BIND")}


def cpu_config0(budget_s=8.0):
    """BASELINE.json configs[0]: the reference's OWN CPU benchmark -- FastVGICP (OpenMP), bundled pair, voxel resolution 1.0, DIRECT1, k = 20,
    100times_reuse loop (README.md:126-128: 4408 ms on one thread = 22.7 registrations/s, 806.53 ms on the 16 threads of an i9-9900K = 124.0).
    Timed here on the oracle (the fp64 restatement of that class) on THIS box's host cores: the one published CPU number the oracle's speed
    can be held against."""
    from fast_gicp_amd import preprocess
    tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
    n = os.cpu_count() or 1
    counts = sorted(set(t for t in (1, 8, 16, 32, 64) if t <= n))
    r = cpu_baseline_vgicp(tgt, src, 1.0, "DIRECT1", "knn", budget_s, mode="reuse", counts=counts)
    sweep = r["thread_sweep_registrations_per_sec"]
    r.update({"config": "BASELINE.json configs[0]: FastVGICP (CPU/OpenMP) on the bundled pair, voxel_res 1.0, DIRECT1, 100times_reuse (oracle = fp64 restatement of the reference class)",
              "published_reference": {"registrations_per_sec_16_threads": 124.0, "registrations_per_sec_1_thread": 22.7, "hardware": "Core i9-9900K (8C/16T)", "source": "README.md:126-128"},
              "this_box_at_16_threads": sweep.get(16), "this_box_at_1_thread": sweep.get(1)})
    return r


def sharded_headline(args, dist, rank, world, local_rank, dev, small=False):
    """--gpus N > 1: the line's `value`. ONE registration stream whose every registration is spread over the N GPUs -- BASELINE configs[4]:
    1M-point map <-> 100k-point scan, DIRECT7, res 0.5, source sharded by spatial tile (Morton ranges), target voxel map replicated, the
    32 x f64 normal-equation block summed over the ranks once per cost evaluation (north_star: RCCL all-reduce over xGMI; route "rccl",
    FVH_BENCH_ROUTE=peer selects the in-kernel mailboxes). A step = scan in (host buffer, as a localisation loop receives it), k-NN k = 20,
    PLANE covariances, align. Strong scaling: the work per registration is fixed, N grows. The same step on ONE GPU is timed first on every
    rank (`n1_same_workload`), so that the speed-up can be read off this line alone."""
    import torch
    from fast_gicp_amd import capi, distributed as D, workloads
    torch.cuda.set_device(local_rank)  # (worker thread of run_with_deadline: the current device is per thread)
    route = os.environ.get("FVH_BENCH_ROUTE", "peer" if small else "rccl")
    n_t, n_s, seed, extent, search_name = (40_000, 20_000, 21, 40.0, "DIRECT7") if small else (1_000_000, 100_000, 44, 150.0, "DIRECT7")
    tgt, src, _ = workloads.synthetic_pair(n_t, n_s, seed=seed, extent=extent)
    steps, warmup = args.steps, args.warmup

    def barrier():
        dist.barrier(); torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fresh():
        c = capi.VGICPCore(local_rank)
        c.set_resolution(0.5); c.set_neighbor_search_method(capi.DIRECT7)
        return c

    # ---- the same step on one GPU (every rank, unsharded) ----
    c1 = fresh()
    c1.set_target_cloud(tgt); c1.find_target_neighbors(20); c1.calculate_target_covariances(capi.REG_PLANE); c1.create_target_voxelmap()

    def step1():
        c1.set_source_cloud(src); c1.find_source_neighbors(20); c1.calculate_source_covariances(capi.REG_PLANE)
        return c1.align()
    for _ in range(min(warmup, 5)):
        r1 = step1()
    n1_steps = max(5, min(steps, 40))
    barrier()
    t0 = time.perf_counter()
    for _ in range(n1_steps):
        r1 = step1()
    c1.synchronize()
    ms1 = max_over_ranks(time.perf_counter() - t0) / n1_steps * 1e3
    c1.close()

    # ---- sharded over the ranks ----
    c = fresh()
    sh = D.ShardedVGICP(c, rank, world, dist, collective=route)
    t0 = time.perf_counter()
    if route == "peer":
        sh.attach_peers(max(n_t, n_s), device_index=local_rank)
    else:
        uid = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        sh.init_device_collective(uid[0])
    attach_ms = (time.perf_counter() - t0) * 1e3
    sh.set_target(tgt)
    c.synchronize()

    def step():
        sh.set_source(src)
        return sh.align()
    for _ in range(warmup):
        r = step()
    n_eval = n_launch = 0
    rep_times = []
    for _rep in range(REPEATS):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
            n_eval += r["num_linearize"] + r["num_error_evals"]
            n_launch += r["num_launches"]
        barrier()
        rep_times.append(max_over_ranks(time.perf_counter() - t0))
    elapsed, el_min, el_max = median_of(rep_times)
    n_regs = steps * REPEATS
    # roofline of the dominant kernel (this rank's cost launches, HIP events on the engine's stream), on a few more steps outside the timed region
    roofline = None
    try:
        c.profile_reset(); c.profile_enable(2)
        ne = 0
        for _ in range(5):
            rr = step()
            ne += rr["num_linearize"] + rr["num_error_evals"]
        c.profile_enable(0)
        cost_ms, cost_n = c.profile_get("cost")
        n_tile = -(-n_s // world)
        n_c = c.get_num_correspondences()  # (sharded: the pairs of this rank's tile)
        bytes_eval = n_tile * 48 + n_tile * 7 * 16 + n_c * 52 + 172
        if cost_n:
            achieved = ne * bytes_eval / (cost_ms * 1e-3) / 1e9
            roofline = {"kernel": "cost_kernel<double,VGICP,%s> on this rank's tile (1 / %d of the scan)" % ("persistent" if cost_n == 5 else "per-transition", world), "bound": "hbm",
                        "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": None,
                        "algorithmic_bytes_per_launch": int(bytes_eval * ne / cost_n), "algorithmic_bytes_per_evaluation": bytes_eval, "evaluations": ne, "launches": cost_n, "avg_launch_us": round(cost_ms / cost_n * 1e3, 3),
                        "note": "rank 0; SURVEY 8(d) B_eval over the tile; no PMC pass exists for the sharded launches (traffic null)"}
    except Exception as ex:  # noqa: BLE001
        roofline = {"error": repr(ex)}
    out = {
        "metric": "registrations/sec (one registration stream, every registration sharded over the GPUs by spatial tile; BASELINE configs[4])",
        "value": round(steps / elapsed, 3), "unit": "registrations/sec", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 5),
        "repeats": REPEATS, "ms_per_step_min": round(el_min / steps * 1e3, 5), "ms_per_step_max": round(el_max / steps * 1e3, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "synthetic %s-point map <-> %s-point scan, seed %d: scan-to-map step (host scan in, k-NN k = 20, PLANE covariances, align)" % (
                       "{:,}".format(n_t), "{:,}".format(n_s), seed),
                   "method": "VGICP", "neighbor_search": search_name, "k_correspondences": 20, "covariance": "knn", "regularization": "PLANE", "voxel_resolution": 0.5,
                   "parallelism": "source sharded by spatial tile (Morton ranges) over %d GPUs, target voxel map replicated; %s" % (world, sh.collective_description()),
                   "route": route},
        "n1_same_workload": {"value": round(1e3 / ms1, 3), "unit": "registrations/sec", "ms_per_step": round(ms1, 5), "steps": n1_steps, "converged": bool(r1["converged"]),
                             "note": "the same step, unsharded, on one GPU (max over the ranks, each on its own GPU): value / this = the strong-scaling speed-up"},
        "speedup_vs_one_gpu": round(ms1 / (elapsed / steps * 1e3), 3),
        "per_registration": {"cost_evaluations": n_eval / n_regs, "kernel_launches_lm": n_launch / n_regs, "converged": bool(r["converged"]),
                             "pose_equals_single_gpu": bool(np.abs(r["T"] - r1["T"]).max() < 1e-9), "max_abs_pose_difference": float(np.abs(r["T"] - r1["T"]).max())},
        "attach_ms": round(attach_ms, 2), "roofline": roofline,
        "cpu_baseline": None,  # (N = 1 only: the N = 1 line carries it)
    }
    dist.barrier()
    if route == "peer":
        c.peer_detach()
    else:
        c.comm_destroy()
    c.close()
    return out


def sharded_leg(args, dist, rank, world, local_rank, dev, small=False):
    """north_star: "large scans shard by spatial tile across up to 8 GPUs with an RCCL all-reduce of the 6x6 / 6x1 normal equations per
    iteration ... 100 k / 1 M synthetic clouds at 1 / 2 / 4 / 8 GPUs". ONE registration spread over the ranks, for both large configurations
    (BASELINE configs[4]: 1M-point map <-> 100k-point scan, DIRECT7; configs[2]-sized 100k <-> 100k, DIRECT27) and BOTH exchange routes:
      "peer"  the engine shards k-NN / covariances / cost evaluation by Morton tile internally, sums meet in peer-mapped mailboxes INSIDE the
              persistent LM kernel (fvh_vgicp_peer_*; a self-check of the mailboxes runs at attach time);
      "rccl"  every rank uploads its spatial tile of the source, ncclAllReduce(32 x f64) on the engine stream after every cost launch.
    Timed per route: the scan-to-map step (source in, k-NN, covariances, align) as two stages, next to the same step on one GPU."""
    import torch
    from fast_gicp_amd import capi, distributed as D, workloads
    torch.cuda.set_device(local_rank)  # this leg runs on a worker thread (run_with_deadline): the current device is per thread
    steps = 20
    out = {"world_size": world, "note": "no scaling curve has been measured on distinct GPUs before this run: the 8-GPU node is the driver's"}

    def barrier():
        dist.barrier(); torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(prepare_source, align, sync):
        """(ms per step, ms of the source stage, ms of the align stage, last result) -- max over the ranks"""
        prepare_source(); r = align(); sync()
        barrier()
        t_src = t_al = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            ta = time.perf_counter()
            prepare_source(); sync()
            tb = time.perf_counter()
            r = align(); sync()
            t_src += tb - ta
            t_al += time.perf_counter() - tb
        dist.barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        return el / steps * 1e3, max_over_ranks(t_src) / steps * 1e3, max_over_ranks(t_al) / steps * 1e3, r

    cases = [("map1m_scan100k", (1_000_000, 100_000, 44, 150.0), capi.DIRECT7, "synthetic 1M-point map <-> 100k-point scan, DIRECT7, res 0.5 (BASELINE configs[4])"),
             ("synth100k", (100_000, 100_000, 42, 60.0), capi.DIRECT27, "synthetic 100k <-> 100k, DIRECT27, res 0.5")]
    routes = ("peer", "rccl")
    if small:  # test-only (tests/test_gpu_bench_flow.py): the leg's control flow with all ranks on ONE GPU -- small clouds, and no RCCL (it needs a GPU per rank)
        cases = [("small", (40_000, 20_000, 21, 40.0), capi.DIRECT7, "synthetic 40k <-> 20k, DIRECT7, res 0.5 (control-flow test)")]
        routes, steps = ("peer",), 3
    for name, (n_t, n_s, seed, extent), search, desc in cases:
        res = {"workload": desc + ": scan-to-map step (host scan in, k-NN k = 20, PLANE covariances, align) over %d GPUs, replicated target voxel map" % world}
        try:
            tgt, src, _ = workloads.synthetic_pair(n_t, n_s, seed=seed, extent=extent)

            def fresh():
                c = capi.VGICPCore(local_rank)
                c.set_resolution(0.5); c.set_neighbor_search_method(search)
                return c

            # ---- one GPU (every rank does the same, unsharded) ----
            c1 = fresh()
            c1.set_target_cloud(tgt); c1.find_target_neighbors(20); c1.calculate_target_covariances(capi.REG_PLANE); c1.create_target_voxelmap()

            def src1():
                c1.set_source_cloud(src); c1.find_source_neighbors(20); c1.calculate_source_covariances(capi.REG_PLANE)
            ms1, ms1_src, ms1_al, r1 = timed(src1, c1.align, c1.synchronize)
            c1.close()
            res["single_gpu"] = {"ms_per_registration": round(ms1, 4), "source_stage_ms": round(ms1_src, 4), "align_ms": round(ms1_al, 4), "converged": bool(r1["converged"])}
            for coll in routes:
                try:
                    c = fresh()
                    sh = D.ShardedVGICP(c, rank, world, dist, collective=coll)
                    t0 = time.perf_counter()
                    if coll == "peer":
                        sh.attach_peers(max(n_t, n_s), device_index=local_rank)   # (runs the mailbox self-check across the devices)
                    else:
                        uid = [capi.comm_unique_id() if rank == 0 else None]
                        dist.broadcast_object_list(uid, src=0)
                        sh.init_device_collective(uid[0])
                    attach_ms = (time.perf_counter() - t0) * 1e3
                    t0 = time.perf_counter()
                    sh.set_target(tgt)
                    c.synchronize()
                    map_ms = (time.perf_counter() - t0) * 1e3
                    ms, ms_src, ms_al, r = timed(lambda: sh.set_source(src), sh.align, c.synchronize)
                    res[coll] = {"registrations_per_sec": round(1e3 / ms, 3), "ms_per_registration": round(ms, 4), "source_stage_ms": round(ms_src, 4), "align_ms": round(ms_al, 4),
                                 "speedup_vs_single_gpu": round(ms1 / ms, 3), "align_speedup_vs_single_gpu": round(ms1_al / ms_al, 3),
                                 "evaluations_per_align": r["num_linearize"] + r["num_error_evals"], "kernel_launches_lm": r["num_launches"],
                                 "attach_ms": round(attach_ms, 2), "sharded_map_preparation_ms": round(map_ms, 2), "converged": bool(r["converged"]),
                                 "pose_equals_single_gpu": bool(np.abs(r["T"] - r1["T"]).max() < 1e-9), "max_abs_pose_difference": float(np.abs(r["T"] - r1["T"]).max()),
                                 "collective": sh.collective_description()}
                    dist.barrier()
                    if coll == "peer":
                        c.peer_detach()
                    else:
                        c.comm_destroy()
                    c.close()
                except Exception as e:  # one route failing must not hide the other
                    res[coll] = {"error": repr(e)}
        except Exception as e:  # the headline number must not depend on this leg
            res["error"] = repr(e)
        out[name] = res
    return out


def concurrent_leg(args, local_rank, d_clouds, n_pts, res, search, K):
    """Aggregate throughput of S independent registration streams on ONE GPU (one engine handle + HIP stream + host thread
    each). The headline `value` is the reference's sequential loop; this shows how much of the chip that loop leaves idle."""
    import threading
    from fast_gicp_amd import capi
    S, steps = args.streams, max(20, args.steps // 2)
    # The oracle's OpenMP runtime (cpu_baseline, OMP_PROC_BIND=close) binds the thread that opened its parallel regions -- this one -- to
    # its first place, and threads started from here inherit that mask: all S host threads would spin on ONE core (round 3's x1.03 at four
    # streams was mostly this). Give the process its CPUs back before starting them.
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except (AttributeError, OSError):
        pass
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
    aborts_loop = sum(c.debug_persist_aborts() for c in cores)
    # the LM loop alone (clouds, covariances and maps stay): what the persistent kernels of S handles make of the chip between them
    def aligns(c, n):
        for _ in range(n):
            c.align()
    na = max(40, steps)
    aligns(cores[0], 5)
    t0 = time.perf_counter()
    aligns(cores[0], na)
    one = na / (time.perf_counter() - t0)
    time.sleep(0.05)
    threads = [threading.Thread(target=aligns, args=(c, na)) for c in cores]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    many = S * na / (time.perf_counter() - t0)
    aborts = sum(c.debug_persist_aborts() for c in cores)
    for c in cores:
        c.close()
    # ... and the same align-only measurement in a process of its own (no torch context, fresh CPU mask): fast_gicp_amd/concurrency_probe.py
    clean = None
    try:
        import subprocess
        p = subprocess.run([sys.executable, "-m", "fast_gicp_amd.concurrency_probe", "1", str(S)], capture_output=True, text=True, timeout=120, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        clean = json.loads(line[-1]) if line else {"error": p.stderr[-300:]}
    except Exception as ex:  # noqa: BLE001
        clean = {"error": repr(ex)}
    return {"streams": S, "steps_per_stream": steps, "registrations_per_sec": round(S * steps / el, 3), "note": "independent registrations, not the sequential reference loop",
            "persistent_launches_aborted_by_watchdog": aborts_loop, "xcd_local_wanted_and_placement_aborts": list(capi.debug_xcd_local()),
            "align_only": {"one_handle_aligns_per_sec": round(one, 1), "all_handles_aligns_per_sec": round(many, 1), "ratio": round(many / one, 3), "persistent_launches_aborted_by_watchdog": aborts - aborts_loop, "in_a_process_without_torch": clean,
                           "note": "align() alone on prepared handles: the co-resident workgroup slots are split between the concurrent persistent LM kernels (SlotPool); the full loop "
                                   "above adds each stream's sort / k-NN / covariance kernels, which wait for room beside the resident LM workgroups"}}


def run_stream(args, steps, warmup, cpu_loops=15, frames_n=10):
    """BASELINE.json configs[3] (KITTI-style streaming, NDT D2D): the loop of src/kitti.cpp:95-128 with every stage on the
    device -- raw ~118k-point frame (resident in HBM) -> ApproximateVoxelGrid 0.25 (kitti.cpp:80-82) -> setInputSource ->
    align (NDTCuda defaults: D2D, DIRECT7, resolution 1.0) -> swapSourceAndTarget.  KITTI is not available offline: frames
    come from the 64-ring LiDAR simulator in fast_gicp_amd/workloads.py (1 m ego-motion per frame); the sequence is walked back
    and forth so consecutive frames are always neighbours.  Single GPU only (a stream of dependent frames does not shard)."""
    import torch
    from fast_gicp_amd import capi, workloads
    F = frames_n
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
        frames = [workloads.lidar_frame(i) for i in range(F)]
    gpu = torch.device("cuda", 0)
    d_frames = [torch.from_numpy(f).to(gpu).contiguous() for f in frames]
    vg, ndt = capi.VoxelGrid(0), capi.NDTCore(0)
    ndt.set_distance_mode(capi.NDT_D2D); ndt.set_neighbor_search_method(capi.DIRECT7); ndt.set_resolution(1.0)
    ndt.set_precision(capi.COMPUTE_FP32 if args.precision == "fp32" else capi.COMPUTE_FP64)
    seq = list(range(F)) + list(range(F - 2, 0, -1))  # 0..F-1..1, repeated

    ptr, n = vg.filter_device(d_frames[0].data_ptr(), len(frames[0]), 0.25, vg.APPROXIMATE)
    ndt.set_target_cloud_device(ptr, n, 3)
    state = {"k": 0, "last": None, "n_ds": 0}

    # the filter runs on the registration handle's stream and reports its count one kernel early (fvh_voxelgrid_filter_device_async):
    # upload, map build and LM kernel are queued behind its last kernel while that kernel runs
    shared = os.environ.get("FVH_BENCH_STREAM_SYNC", "0") != "1"  # (A/B knob: 1 = the filter on its own stream, synchronous count)
    TAKE = os.environ.get("FVH_BENCH_TAKE_FILTER_OUTPUT", "1") != "0"  # (A/B knob: 0 = device pointer + widening kernel, as before round 6's last day)
    if shared:
        vg.share_stream(ndt)

    def step():
        state["k"] += 1
        i = seq[state["k"] % len(seq)]
        ptr, n = vg.filter_device(d_frames[i].data_ptr(), len(frames[i]), 0.25, vg.APPROXIMATE, asynchronous=shared, want_pointer=not TAKE)
        if TAKE:
            ndt.set_source_cloud_from_voxelgrid(vg)  # the filter's float4 output IS the cloud: no widening kernel (fvh_ndt_set_source_cloud_from_voxelgrid)
        else:
            ndt.set_source_cloud_device(ptr, n, 3)
        state["last"] = ndt.align()
        ndt.swap_source_and_target()
        state["n_ds"] += n

    for _ in range(warmup):
        step()
    # accuracy on the first lap (not timed): per-frame relative pose vs the simulator's ground truth
    errs = []
    state["k"] = 0
    ptr, n = vg.filter_device(d_frames[0].data_ptr(), len(frames[0]), 0.25, vg.APPROXIMATE)
    ndt.set_target_cloud_device(ptr, n, 3)
    for i in range(1, F):
        step()
        if not kitti_dir:  # ground truth exists for the simulator only
            gt = np.linalg.inv(workloads.lidar_pose(i - 1)) @ workloads.lidar_pose(i)
            errs.append(workloads.pose_error(gt, state["last"]["T"])[0])
    state["k"] = F - 1
    profile = not args.no_profile
    ndt.profile_reset(); vg.profile_reset()
    ndt.profile_enable(False); vg.profile_enable(False)
    state["n_ds"] = 0
    n_eval = n_launch = 0
    rep_times = []
    for _rep in range(REPEATS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(steps):
            if profile and it % 9 <= 1:
                ndt.profile_enable(2 if it % 9 == 0 else 0)  # the LM kernel only (two events per sampled frame); map build and filter are timed after the timed region
            step()
            n_eval += state["last"]["num_linearize"] + state["last"]["num_error_evals"]
            n_launch += state["last"]["num_launches"]
        ndt.synchronize(); torch.cuda.synchronize()
        rep_times.append(time.perf_counter() - t0)
        ndt.profile_enable(False)
    elapsed, el_min, el_max = median_of(rep_times)
    n_regs = steps * REPEATS
    ndt.profile_enable(False); vg.profile_enable(False)
    stage_ms, roofline, roofline_ds = {}, None, None
    cost_sample = ndt.profile_get("cost") if profile else (0.0, 0)
    n_ds = state["n_ds"] // max(n_regs, 1)
    if profile:  # the other stages: 18 more frames of the same loop, every kernel class bracketed, outside the timed region
        ndt.profile_reset(); vg.profile_reset()
        ndt.profile_enable(1); vg.profile_enable(True)
        for _ in range(18):
            step()
        ndt.profile_enable(False); vg.profile_enable(False)
    n_raw = int(np.mean([len(f) for f in frames]))
    # sizes of one registration for the byte model (outside the timed region: one more frame, no swap afterwards)
    i = seq[(state["k"] + 1) % len(seq)]
    ptr, n = vg.filter_device(d_frames[i].data_ptr(), len(frames[i]), 0.25, vg.APPROXIMATE)
    ndt.set_source_cloud_device(ptr, n, 3)
    ndt.align()
    n_c = ndt.get_num_correspondences()
    n_sv = ndt.get_num_voxels("source")
    n_tv = ndt.get_num_voxels("target")
    ndt.swap_source_and_target()
    state["k"] += 1
    if profile:
        if cost_sample[1]:
            stage_ms["cost"] = {"total_ms": round(cost_sample[0], 3), "launches": cost_sample[1], "avg_us": round(cost_sample[0] / cost_sample[1] * 1e3, 3)}
        ms, n = ndt.profile_get("voxelmap")
        if n:
            stage_ms["voxelmap"] = {"total_ms": round(ms, 3), "launches": n, "avg_us": round(ms / n * 1e3, 3), "sampled": "after the timed region"}
        if "cost" in stage_ms:
            # the DOMINANT kernel of this loop: the LM kernel's NDT D2D instantiation. SURVEY 8(d) with the source voxels as the source
            # elements: B_eval = N_sv * 48 + N_sv * N_off * 16 + N_c * 52 + 172 per evaluation; one persistent launch runs them all
            ms, n = cost_sample
            evals = n_eval / max(n_launch, 1)
            bytes_eval = n_sv * 48 + n_sv * 7 * 16 + n_c * 52 + 172
            b = bytes_eval * evals
            ach = b / (ms / n * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic("lidar_stream_cost") if args.precision == "fp64" else (None, "PMC passes exist for the fp64 kernel only")
            sq = ((pmc_entry("lidar_stream_cost")[0] or {}).get("sq") or {}) if traffic is not None else {}
            roofline = {"kernel": "cost_kernel<%s,NDT_D2D,persistent>" % ("double" if args.precision == "fp64" else "float"), "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(ach / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_src,
                        "valu_busy_frac_of_wave_cycles": sq.get("valu_busy_frac_of_wave_cycles"), "valu_issue_utilisation": sq.get("valu_issue_utilisation"),
                        "algorithmic_bytes_per_launch": int(b), "algorithmic_bytes_per_evaluation": int(bytes_eval), "evaluations_per_launch": round(evals, 3),
                        "source_voxels": n_sv, "target_voxels": n_tv, "correspondences": n_c, "avg_launch_us": round(ms / n * 1e3, 3), "launches": n,
                        "note": "a few thousand source voxels x 7 offsets: ~%d trips of a barrier-separated latency chain (per trip ~5 us of one-item-per-thread main loop "
                                "+ ~6 us of hand-offs and LM step, tools/persist_timing.py --ndt); the working set (~1 MB) never leaves the caches" % round(evals / 2 + 1)}
        ms, n = vg.profile_get()
        if n:
            stage_ms["downsample"] = {"total_ms": round(ms, 3), "launches": n, "avg_us": round(ms / n * 1e3, 3)}
            # second entry: the stage with real bytes in this loop, the filter. Algorithmic bytes: read N x 12 B, write M x 12 B
            b = n_raw * 12 + n_ds * 12
            ach = b / (ms / n * 1e-3) / 1e9
            # PMC traffic of the chain = the sum over its four kernels (one FETCH_SIZE and one WRITE_SIZE pass of this command, profiles/r06_pmc.json)
            parts = [pmc_traffic("lidar_stream_downsample_" + k)[0] for k in ("hist", "scatter", "mark", "emit")]
            ds_traffic = int(sum(parts)) if all(p is not None for p in parts) else None
            roofline_ds = {"kernel": "voxel-grid filter (ApproximateVoxelGrid, 4 launches: slot histogram, scatter, mark, emit; the two scans are recomputed per workgroup)", "bound": "hbm", "achieved": round(ach, 2),
                           "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5), "traffic": ds_traffic, "traffic_per_kernel": dict(zip(("hist", "scatter", "mark", "emit"), parts)), "algorithmic_bytes_per_launch": b,
                           "avg_launch_us": round(ms / n * 1e3, 3), "launches": n,
                           "note": "1.4 MB of input per frame: launch/latency bound (a chain of four dependent small kernels), not HBM bound; `traffic` = the four kernels' PMC bytes summed"}
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        best = None
        sweep = {}
        for th in thread_counts():
            g = O.NDT(threads=th, mode=O.D2D, search=O.DIRECT7)
            loops = max(4, cpu_loops // 3)
            g.set_target(O.approx_voxelgrid(frames[0], 0.25))
            t1 = time.perf_counter()
            for k in range(1, loops + 1):
                g.set_source(O.approx_voxelgrid(frames[seq[k % len(seq)]], 0.25))
                g.align()
                g.swap()
            rate = loops / (time.perf_counter() - t1)
            sweep[th] = round(rate, 3)
            if best is None or rate > best[1]:
                best = (th, rate)
        cpu = {"value": best[1] if best else None, "unit": "registrations/sec", "cores": best[0], "kind": "port",
               "sample": "%d frames of the same loop per thread count: oracle ApproximateVoxelGrid (1 thread, as PCL) + oracle NDT D2D (OpenMP); best of the sweep reported" % max(4, cpu_loops // 3),
               "thread_sweep_registrations_per_sec": sweep, "host_cores": os.cpu_count()}
        cpu["value"] = round(cpu["value"], 3)
    # ---- side leg: the same loop as a two-stage pipeline on ONE host thread (fvh_ndt_align_async / _wait + the handle's prepared-source
    # slot): while the LM kernel of frame k runs on the main stream, frame k+1 is filtered, widened and its voxel map built on the handle's
    # second stream. The odometry chain itself stays sequential; poses are bit-identical to the plain loop (tests/test_gpu_streaming.py).
    # Not `value`: kitti.cpp is a sequential loop.
    pipelined = None
    try:
        if getattr(args, "no_pipelined_leg", False):
            raise RuntimeError("skipped (--no-pipelined-leg)")
        vg.share_stream(None)
        vg.share_prepare_stream(ndt)
        P = max(steps, 20)
        order = [seq[(k + 1) % len(seq)] for k in range(P + 1)]
        ptrs = [t.data_ptr() for t in d_frames]
        lens = [len(f) for f in frames]

        def prepare(i):
            ptr, n = vg.filter_device(ptrs[i], lens[i], 0.25, vg.APPROXIMATE, asynchronous=True, want_pointer=not TAKE)
            if TAKE:
                ndt.prepare_source_from_voxelgrid(vg)
            else:
                ndt.prepare_source_device(ptr, n, 3)

        def run(count):
            r = None
            for k in range(count):
                ndt.adopt_prepared_source()
                ndt.align_async()
                prepare(order[k + 1])
                r = ndt.align_wait()
                ndt.swap_source_and_target()
            return r

        ptr, n = vg.filter_device(ptrs[seq[0]], lens[seq[0]], 0.25, vg.APPROXIMATE)
        ndt.set_target_cloud_device(ptr, n, 3)
        prepare(order[0])
        run(5)
        times = []
        for _rep in range(REPEATS):
            ndt.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = run(P)
            ndt.synchronize(); torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        el, el_lo, el_hi = median_of(times)
        pipelined = {"registrations_per_sec": round(P / el, 3), "ms_per_step": round(el / P * 1e3, 5), "ms_per_step_min": round(el_lo / P * 1e3, 5), "ms_per_step_max": round(el_hi / P * 1e3, 5),
                     "steps": P, "repeats": REPEATS, "converged": bool(r["converged"]), "kernel_launches_lm": r["num_launches"],
                     "note": "one host thread: adopt_prepared_source, align_async, [filter + prepare_source_device of frame k+1 on the handle's second stream], align_wait, swap"}
    except Exception as ex:  # noqa: BLE001
        pipelined = {"error": repr(ex)}
    vg.close(); ndt.close()
    return {"pipelined": pipelined, "metric": "registrations/sec (frame-by-frame odometry, kitti.cpp loop incl. downsampling)", "value": round(steps / elapsed, 3), "unit": "registrations/sec", "n_gpus": 1,
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 5), "repeats": REPEATS, "ms_per_step_min": round(el_min / steps * 1e3, 5),
            "ms_per_step_max": round(el_max / steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.precision == "fp64" else "f32", "data": "KITTI" if kitti_dir else "synthetic",
            "config": {"workload": "%s, %d raw pts/frame -> ApproximateVoxelGrid 0.25 -> %d pts; NDT D2D, DIRECT7, res 1.0; %d-frame sequence walked back and forth" % (
                "KITTI frames from FVH_KITTI_DIR" if kitti_dir else "simulated 64-ring LiDAR", n_raw, n_ds, F),
                "method": "NDT_D2D", "neighbor_search": "DIRECT7", "voxel_resolution": 1.0, "parallelism": "single GPU"},
            "per_registration": {"cost_evaluations": n_eval / n_regs, "kernel_launches_lm": n_launch / n_regs, "converged": bool(state["last"]["converged"])},
            "accuracy": ({"max_frame_translation_error_m": round(float(max(errs)), 4), "mean_frame_translation_error_m": round(float(np.mean(errs)), 4)} if errs else None),
            "roofline": roofline, "roofline_downsample": roofline_ds, "cpu_baseline": cpu, "stages": stage_ms}


def run_registration(args, workload, cov, search_name, steps, warmup, local_rank=0, dist=None, dev=None, world=1, rank=0, headline=False, cpu_budget=12.0, cpu=True):
    """The VGICP loop of one workload on this rank's GPU; returns the result dictionary (rank 0) or None."""
    import torch
    from fast_gicp_amd import capi
    tgt, src, res, desc = make_workload(workload)
    search = {"DIRECT27": capi.DIRECT27, "DIRECT7": capi.DIRECT7, "DIRECT1": capi.DIRECT1}[search_name]
    K = 20
    gpu = torch.device("cuda", local_rank)
    d_clouds = [torch.from_numpy(tgt).to(gpu).contiguous(), torch.from_numpy(src).to(gpu).contiguous()]  # inputs resident in HBM
    h_clouds = [tgt, src]
    n_pts = [len(tgt), len(src)]

    core = capi.VGICPCore(local_rank)
    core.set_resolution(res)
    core.set_neighbor_search_method(search)
    core.set_kernel_params(0.5, 2.5)  # align.cpp:210 setKernelWidth(0.5) -> max_dist 2.5
    core.set_precision(capi.COMPUTE_FP32 if args.precision == "fp32" else capi.COMPUTE_FP64)

    kd = None
    if cov == "kdtree":  # FastVGICPCuda's default, NearestNeighborMethod::CPU_PARALLEL_KDTREE (fast_vgicp_cuda_impl.hpp:152-167): neighbours from a host
        import pygicp   # kd-tree (OpenMP), handed to the device, covariances there -- the "vgicp_cuda (parallel_kdtree)" row of align.cpp:190-196
        kd = pygicp._kdtree_knn
    cloud_of = {"target": 0, "source": 1}

    def estimate_cov(which):
        if cov == "knn":
            getattr(core, "find_%s_neighbors" % which)(K)
            getattr(core, "calculate_%s_covariances" % which)(capi.REG_PLANE)
        elif cov == "kdtree":
            idx = kd(h_clouds[cloud_of[which]].astype(np.float64), K)
            getattr(core, "set_%s_neighbors" % which)(K, np.ascontiguousarray(idx, np.int32))
            getattr(core, "calculate_%s_covariances" % which)(capi.REG_PLANE)
        else:
            getattr(core, "calculate_%s_covariances_rbf" % which)(capi.REG_PLANE)

    # "single" (align.cpp:58-68): both clouds from scratch
    d_ptrs = [t.data_ptr() for t in d_clouds]  # (Tensor.data_ptr() is a microsecond of Python on the path between an align and the next launch)
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

    state = {"next": 0, "last": first, "host": False}  # after "single": target = cloud 0, source = cloud 1

    def set_source(i):
        if state["host"]:
            core.set_source_cloud(h_clouds[i])  # align.cpp:94-96: a host cloud per registration (H2D inside)
        else:
            core.set_source_cloud_device(d_ptrs[i], n_pts[i], 3)

    if workload == "synth1m":
        # map-vs-scan localisation (BASELINE configs[4] shape): the 1M-point map stays the target, every step registers a scan
        def step():
            set_source(1)
            cloud_of["source"] = 1
            estimate_cov("source")
            state["last"] = core.align()
            state["next"] = 0
    else:
        def step():
            # reg.swapSourceAndTarget(); reg.clearSource(); reg.setInputTarget(target_) [no-op]; reg.setInputSource(source_); reg.align()
            core.swap_source_and_target()
            i = state["next"]
            set_source(i)
            cloud_of["source"], cloud_of["target"] = i, 1 - i
            estimate_cov("source")
            state["last"] = core.align()
            state["next"] = 1 - i

    for _ in range(warmup):
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

    n_lin = n_err = n_launch = 0
    rep_times = []
    for _rep in range(REPEATS):  # the timed region, REPEATS times: exactly `steps` steps between barrier + synchronize on both sides, max over the ranks
        barrier()
        t0 = time.perf_counter()
        for it in range(steps):
            g = _rep * steps + it  # (counted over all repeats: the samples then alternate between the two directions of the pair whatever `steps` is)
            if profile and g % PROFILE_EVERY <= 1:  # (on at 0, off again at 1: a call per step would sit between an align and the next launch, with the GPU idle)
                core.profile_enable(2 if g % PROFILE_EVERY == 0 else 0)  # level 2: the LM kernel only (the roofline's launch time); the other stages are timed after the timed region
            step()
            n_lin += state["last"]["num_linearize"]
            n_err += state["last"]["num_error_evals"]
            n_launch += state["last"]["num_launches"]
        barrier()
        el = time.perf_counter() - t0
        core.profile_enable(False)
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        rep_times.append(el)
    elapsed, el_min, el_max = median_of(rep_times)
    n_regs = steps * REPEATS  # registrations the per-registration counters were summed over

    # ---- align.cpp:73-83 "100times": clearTarget / clearSource / setInputTarget / setInputSource / align -- BOTH clouds from scratch per registration ----
    full_ms = None
    if world == 1 and workload != "synth1m":
        def full_step():
            core.set_target_cloud_device(d_ptrs[0], n_pts[0], 3)
            cloud_of["target"], cloud_of["source"] = 0, 1
            estimate_cov("target")
            core.create_target_voxelmap()
            core.set_source_cloud_device(d_ptrs[1], n_pts[1], 3)
            estimate_cov("source")
            return core.align()
        for _ in range(2):
            full_step()
        core.synchronize()
        t1 = time.perf_counter()
        n_full = max(10, min(steps, 50))
        for _ in range(n_full):
            state["last"] = full_step()
        core.synchronize()
        full_ms = (time.perf_counter() - t1) / n_full * 1e3
        state["next"] = 0  # (target = cloud 0, source = cloud 1 again)

    # ---- the same loop, host clouds in (PCIe-inclusive; never `value`) ----
    host_leg = None
    if not args.no_host_leg and world == 1:
        state["host"] = True
        for _ in range(2):
            step()
        core.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        core.synchronize()
        el_h = time.perf_counter() - t1
        state["host"] = False
        host_leg = {"value": round(steps / el_h, 3), "unit": "registrations/sec", "ms_per_step": round(el_h / steps * 1e3, 5),
                    "note": "same loop, source cloud handed over as a HOST buffer every registration (fvh_vgicp_set_source_cloud: %d bytes copied into the handle's pinned staging buffer and read from there by the widening kernel over PCIe), as src/align.cpp:94-96 does"
                            % (n_pts[1] * 12)}
    # ---- the same loop as a two-stage pipeline: the next source (sort, k-NN, covariances, its voxel map) is prepared on the handle's
    #      second stream beside the running LM kernel (fvh_vgicp_prepare_source_device / align_async / align_wait); never `value` ----
    pipelined_leg = None
    if not args.no_pipelined_leg and world == 1 and workload != "synth1m" and cov in ("knn", "rbf"):
        rbf = cov == "rbf"
        PSTAGES = int(os.environ.get("BENCH_PIPE_STAGES", "2"))
        seq_T = {}
        for _ in range(2):  # what the sequential loop finds, per direction of the pair
            step()
            seq_T[1 - state["next"]] = state["last"]["T"]

        def pstep():
            i = state["next"]  # the cloud that becomes the source next = the one that is the target now
            core.align_async()  # (first: with the preparation queued BEFORE the LM kernel the kernel's dispatch waits behind the chain's first kernels -- 7,020 -> 6,580)
            core.prepare_source_device(d_ptrs[i], n_pts[i], 3, K, capi.REG_PLANE, rbf, PSTAGES)
            state["last"] = core.align_wait()
            core.swap_source_and_target()
            core.adopt_prepared_source()
            if PSTAGES < 2:
                cloud_of["source"] = i
                (core.calculate_source_covariances_rbf if rbf else core.calculate_source_covariances)(capi.REG_PLANE)
            state["next"] = 1 - i
        for _ in range(max(4, warmup)):
            pstep()
        core.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            pstep()
        core.synchronize()
        el_p = time.perf_counter() - t1
        dev_T = max(float(np.abs(state["last"]["T"] - seq_T[state["next"]]).max()), 0.0)  # (the align that just finished registered cloud `next` as the source)
        pipelined_leg = {"value": round(steps / el_p, 3), "unit": "registrations/sec", "ms_per_step": round(el_p / steps * 1e3, 5),
                         "max_abs_pose_difference_to_the_sequential_loop": dev_T, "persist_aborts": core.debug_persist_aborts(),
                         "prepared_stages": PSTAGES,
                         "note": "the same work per registration as `value` (one cloud sorted / searched / covariances, one voxel map, one align): the preparation of scan k+1 "
                                 "(fvh_vgicp_prepare_source_device, stages: 1 order + neighbours, 2 + covariances, 3 + its voxel map) runs on the handle's second stream "
                                 "beside the LM kernel of scan k; one host thread. Never `value`."}
        # back to the sequential state the legs below expect: target = cloud 0 (map built), source = cloud 1
        core.set_target_cloud_device(d_ptrs[0], n_pts[0], 3)
        cloud_of["target"], cloud_of["source"] = 0, 1
        estimate_cov("target")
        core.create_target_voxelmap()
        core.set_source_cloud_device(d_ptrs[1], n_pts[1], 3)
        estimate_cov("source")
        state["last"] = core.align()
        state["next"] = 0
    if rank != 0:
        core.close()
        return None

    # ---- roofline of the dominant kernel class of the LM loop (cost evaluation) ----
    n_off = {"DIRECT27": 27, "DIRECT7": 7, "DIRECT1": 1}[search_name]
    n_c = core.get_num_correspondences()  # valid (source, voxel) pairs of the last linearisation
    n_src = n_pts[1 - state["next"]]
    # B_eval = N_s*48 + N_s*N_off*16 + N_c*52 (+172 B out)
    bytes_eval = n_src * 48 + n_src * n_off * 16 + n_c * 52 + 172
    roofline = None
    stage_ms = {}
    if profile:
        cost_ms, cost_n = core.profile_get("cost")  # sampled INSIDE the timed region (every PROFILE_EVERY-th registration, two events each)
        if cost_n:
            stage_ms["cost"] = {"total_ms": round(cost_ms, 3), "launches": cost_n, "avg_us": round(cost_ms / cost_n * 1e3, 3)}
        # the other stages: 18 more registrations of the same loop AFTER the timed region, every kernel class bracketed (twelve event
        # records per registration cost it ~25 %: they no longer sit in the number this line reports)
        core.profile_reset(); core.profile_enable(1)
        for _ in range(18):
            step()
        core.profile_enable(0)
        for cls in ("knn", "cov", "rbf", "voxelmap", "sort"):
            ms, n = core.profile_get(cls)
            if n:
                stage_ms[cls] = {"total_ms": round(ms, 3), "launches": n, "avg_us": round(ms / n * 1e3, 3), "sampled": "after the timed region"}
        if cost_n:
            avg_s = cost_ms / cost_n * 1e-3
            # a registration's cost evaluations move (n_lin + n_err) * B_eval algorithmic bytes. One launch of the
            # persistent LM kernel runs ALL of them (kernel_launches_lm == 1); the multi-launch path spreads them over its launches.
            evals_per_launch = (n_lin + n_err) / max(n_launch, 1)
            bytes_launch = bytes_eval * evals_per_launch
            achieved = bytes_launch / avg_s / 1e9
            persistent = n_launch == n_regs
            key = workload + ("_rbf" if cov == "rbf" else "") + "_cost"
            kname = "cost_kernel<%s,VGICP,%s>" % ("double" if args.precision == "fp64" else "float", "persistent" if persistent else "per-transition")
            traffic, traffic_src = pmc_traffic(key) if (persistent and args.precision == "fp64") else (None, "PMC passes exist for the persistent fp64 kernel only")
            sq = ((pmc_entry(key)[0] or {}).get("sq") or {}) if traffic is not None else {}
            roofline = {"kernel": kname, "bound": "hbm", "achieved": round(achieved, 2),
                        "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_src,
                        "valu_busy_frac_of_wave_cycles": sq.get("valu_busy_frac_of_wave_cycles"), "valu_issue_utilisation": sq.get("valu_issue_utilisation"),
                        "algorithmic_bytes_per_launch": int(bytes_launch),
                        "algorithmic_bytes_per_evaluation": bytes_eval, "evaluations_per_launch": round(evals_per_launch, 3),
                        "avg_launch_us": round(avg_s * 1e6, 3), "launches": cost_n,
                        "note": ("17k-point working set (~3 MB) is L2/Infinity-Cache resident and the fused trips share their loads between the trial evaluation and the next "
                                 "linearisation: this kernel is bound by the latency chain of its %d barrier-separated trips, not by HBM; `traffic` (PMC) is what actually reaches HBM"
                                 % round((n_err / n_regs) + 1) if workload == "bundled17k"
                                 else "launch average over the LM launches of the timed region; `traffic` (PMC) is what reached HBM")}
    if cov == "knn" and "knn" in stage_ms:  # VALU-bound stage -> VALU utilisation (PMC) + pair-evaluation rate
        stage_ms["knn"]["roofline"] = valu_roofline(workload + "_knn", "knn_tiled1_kernel", stage_ms["knn"]["avg_us"], n_src, n_src)
    if cov == "rbf" and "rbf" in stage_ms:
        stage_ms["rbf"]["roofline"] = valu_roofline(workload + "_rbf_rbf", "cov_rbf1_kernel (+ cov_rbf_finish_kernel)", stage_ms["rbf"]["avg_us"], n_src, n_src)

    cpu_res = None
    if cpu and not args.no_cpu_baseline and world == 1:
        # (the 100k / 1M configurations prepare their clouds once per thread count: only 16 and 64 threads are tried there)
        counts = thread_counts() if workload == "bundled17k" else [t for t in thread_counts() if t in (16, 64)] or thread_counts()[-1:]
        cpu_res = cpu_baseline_vgicp(tgt, src, res, search_name, cov, cpu_budget, mode="map" if workload == "synth1m" else "reuse", counts=counts)

    conc = None
    if headline and args.streams > 1 and world == 1 and cov == "knn" and workload != "synth1m":
        try:
            conc = concurrent_leg(args, local_rank, d_clouds, n_pts, res, search, K)
        except Exception as ex:
            conc = {"error": repr(ex)}
    total_regs = steps * world
    out = {
        "metric": "registrations/sec (100-iter reuse) + final fitness_score; achieved HBM GB/s",
        "value": round(total_regs / elapsed, 3), "unit": "registrations/sec", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 5),
        "repeats": REPEATS, "ms_per_step_min": round(el_min / steps * 1e3, 5), "ms_per_step_max": round(el_max / steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == "fp64" else "f32",
        "data": "bundled scans (real LiDAR)" if workload == "bundled17k" else "synthetic",
        "config": {"workload": desc, "method": "VGICP", "neighbor_search": search_name, "k_correspondences": K, "covariance": cov, "regularization": "PLANE",
                   "voxel_resolution": res, "parallelism": "1 registration stream per GPU" if world > 1 else "single GPU",
                   "loop": "scan-to-map (map stays the target)" if workload == "synth1m" else "100times_reuse (align.cpp:87-101)", "inputs": "resident in HBM before the timed region"},
        "fitness_score": round(fitness, 6), "single_ms": round(single_ms, 3), "ms_per_reg_100times": None if full_ms is None else round(full_ms, 5),
        "per_registration": {"linearize": n_lin / n_regs, "error_evals": n_err / n_regs, "kernel_launches_lm": n_launch / n_regs, "converged": bool(state["last"]["converged"]),
                             "persistent_launches_aborted_by_watchdog": core.debug_persist_aborts()},
        "host_clouds_in": host_leg,
        "pipelined": pipelined_leg,
        "roofline": roofline, "cpu_baseline": cpu_res, "stages": stage_ms, "profiled_timed_region": ("every %dth registration" % PROFILE_EVERY) if profile else False,
    }
    if conc is not None:
        out["concurrent_streams"] = conc
    core.close()
    return out


def run_fgicp(args, steps, warmup, local_rank=0):
    """`bundled17k_fgicp`: FastGICP -- nearest-target-point correspondences instead of voxel lookups -- on the bundled pair in the
    100times_reuse loop of align.cpp:87-101 (swap, new source: sort + k-NN + covariances, align), and getFitnessScore (the other half of the metric)
    on the result. Both run the exact 1-NN search nn1_rows_kernel: per LM transition in the registration, once in the fitness score."""
    import torch
    from fast_gicp_amd import capi
    tgt, src, res, desc = make_workload("bundled17k")
    gpu = torch.device("cuda", local_rank)
    d_clouds = [torch.from_numpy(tgt).to(gpu).contiguous(), torch.from_numpy(src).to(gpu).contiguous()]
    ptrs, n_pts = [t.data_ptr() for t in d_clouds], [len(tgt), len(src)]
    c = capi.VGICPCore(local_rank)
    c.set_target_cloud_device(ptrs[0], n_pts[0], 3); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE)
    c.set_source_cloud_device(ptrs[1], n_pts[1], 3); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
    r = c.gicp_align()
    T = r["T"].astype(np.float32).astype(np.float64)
    fitness = c.fitness_score(T)
    nxt = [0]

    def step():
        c.gicp_swap_source_and_target()
        i = nxt[0]
        c.set_source_cloud_device(ptrs[i], n_pts[i], 3); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
        out = c.gicp_align()
        nxt[0] = 1 - i
        return out
    for _ in range(warmup):
        r = step()
    times, n_launch = [], 0
    for _rep in range(REPEATS):
        c.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
            n_launch += r["num_launches"]
        c.synchronize(); torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    el, lo, hi = median_of(times)
    # the same loop with the next source prepared on the second stream while the host drives the LM transitions of the current pair
    # (fvh_vgicp_prepare_source_device is legal at any time; FastGICP's LM loop is two small launches per transition: the chip is mostly free)
    pipelined = None
    if not args.no_pipelined_leg:
        seq_T = {}
        for _ in range(2):
            out = step()
            seq_T[1 - nxt[0]] = out["T"]

        def pstep():
            i = nxt[0]
            c.prepare_source_device(ptrs[i], n_pts[i], 3, 20, capi.REG_PLANE, False, 2)
            out = c.gicp_align()
            c.gicp_swap_source_and_target()
            c.adopt_prepared_source()
            nxt[0] = 1 - i
            return out
        # (the sequential step leaves source + target set: one more sequential swap + source, then the pipelined steps take over)
        c.gicp_swap_source_and_target()
        c.set_source_cloud_device(ptrs[nxt[0]], n_pts[nxt[0]], 3); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
        nxt[0] = 1 - nxt[0]
        for _ in range(max(4, warmup)):
            out = pstep()
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = pstep()
        c.synchronize()
        el_p = time.perf_counter() - t0
        # (the pose of the last align against the sequential loop's pose of the same direction of the pair)
        diff = min(float(np.abs(out["T"] - seq_T[k]).max()) for k in seq_T)
        pipelined = {"value": round(steps / el_p, 3), "unit": "registrations/sec", "ms_per_step": round(el_p / steps * 1e3, 5), "max_abs_pose_difference_to_the_sequential_loop": diff}
        # back to the sequential state for the profiled steps below
        c.gicp_swap_source_and_target()
        c.set_source_cloud_device(ptrs[nxt[0]], n_pts[nxt[0]], 3); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
        c.gicp_align()
        nxt[0] = 1 - nxt[0]
    stage = {}
    if not args.no_profile:
        c.profile_reset(); c.profile_enable(1)
        for _ in range(10):
            step()
        for _ in range(10):
            c.fitness_score(T)
        c.profile_enable(0)
        for cls in ("gicp_nn", "cost", "knn", "sort", "cov", "fitness"):
            ms, n = c.profile_get(cls)
            if n:
                stage[cls] = {"total_ms": round(ms, 3), "launches": n, "avg_us": round(ms / n * 1e3, 3)}
    roof = None
    if "gicp_nn" in stage:
        roof = valu_roofline("fgicp17k_nn1", "nn1_rows_kernel", stage["gicp_nn"]["avg_us"], n_pts[1], n_pts[0])
        roof["avg_launch_us"] = stage["gicp_nn"]["avg_us"]
        roof["note"] = "four queries per wave (one per 16-lane row), boxes nearest first; the full-sweep-equivalent rate is what a brute-force N x N kernel would have to sustain"
    c.close()
    return {"metric": "registrations/sec (100-iter reuse), FastGICP (nearest-point correspondences)", "value": round(steps / el, 3), "unit": "registrations/sec", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "repeats": REPEATS, "ms_per_step": round(el / steps * 1e3, 5), "ms_per_step_min": round(lo / steps * 1e3, 5), "ms_per_step_max": round(hi / steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "bundled scans (real LiDAR)", "fitness_score": round(float(fitness), 6),
            "fitness_us": stage.get("fitness", {}).get("avg_us"),
            "config": {"workload": desc, "method": "FastGICP (fast_gicp_impl.hpp:118-240 on the device)", "k_correspondences": 20, "covariance": "knn", "regularization": "PLANE",
                       "loop": "100times_reuse (align.cpp:87-101)", "inputs": "resident in HBM before the timed region"},
            "per_registration": {"kernel_launches_lm": n_launch / (steps * REPEATS), "converged": bool(r["converged"])},
            "pipelined": pipelined, "roofline": roof, "stages": stage}


def run_reference_api(args, steps, warmup):
    """`bundled17k_parallel_kdtree`: what a user of the reference gets who switches libraries and changes NOTHING -- pygicp.FastVGICPCuda() with its default
    NearestNeighborMethod::CPU_PARALLEL_KDTREE (fast_vgicp_cuda_impl.hpp:27), host clouds in, the 100times_reuse loop of align.cpp:87-101 through the
    reference-API C++ classes (include/fast_gicp_amd/registration.hpp). That enum value asks for exact k-NN lists; round 6 serves it from the device
    search (identical lists), the host kd-tree stays behind set_host_kdtree(True): both are timed."""
    import pygicp
    tgt, src, res, desc = make_workload("bundled17k")
    clouds = [tgt.astype(np.float64), src.astype(np.float64)]

    def loop(host_tree, steps, warmup):
        reg = pygicp.FastVGICPCuda()
        reg.set_resolution(res)
        reg.set_neighbor_search_method("DIRECT27")
        if host_tree:
            reg.set_host_kdtree(True)
        t0 = time.perf_counter()
        reg.set_input_target(clouds[0]); reg.set_input_source(clouds[1])
        T = reg.align()
        single_ms = (time.perf_counter() - t0) * 1e3
        fitness = reg.get_fitness_score()
        nxt = [0]

        def step():
            # reg.swapSourceAndTarget(); reg.clearSource(); reg.setInputTarget(target) [the same cloud: a no-op in C++]; reg.setInputSource(source); reg.align()
            reg.swap_source_and_target(); reg.clear_source()
            reg.set_input_source(clouds[nxt[0]])
            reg.align()
            nxt[0] = 1 - nxt[0]
        for _ in range(warmup):
            step()
        times = []
        for _rep in range(REPEATS):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            times.append(time.perf_counter() - t0)
        el, lo, hi = median_of(times)
        return {"value": round(steps / el, 3), "ms_per_step": round(el / steps * 1e3, 5), "ms_per_step_min": round(lo / steps * 1e3, 5), "ms_per_step_max": round(hi / steps * 1e3, 5),
                "single_ms": round(single_ms, 3), "fitness_score": round(float(fitness), 6), "converged": bool(reg.has_converged())}

    def pipelined_loop(steps, warmup):
        # the same registrations through the pipeline calls of the class (not in the reference): the numpy conversion, the staging copy and the device's
        # sort / k-NN / covariances of the NEXT scan run while the LM kernel of the current pair does
        reg = pygicp.FastVGICPCuda()
        reg.set_resolution(res)
        reg.set_neighbor_search_method("DIRECT27")
        reg.set_input_target(clouds[0]); reg.set_input_source(clouds[1])
        T_seq = {1: reg.align()}
        reg.swap_source_and_target(); reg.clear_source(); reg.set_input_source(clouds[0])
        T_seq[0] = reg.align()
        nxt = [1]
        last = [None]

        def step():
            reg.align_async()
            reg.prepare_next_source(clouds[nxt[0]])
            last[0] = reg.align_wait()
            reg.swap_source_and_target()
            reg.adopt_prepared_source()
            nxt[0] = 1 - nxt[0]
        for _ in range(max(warmup, 4)):
            step()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        el = time.perf_counter() - t0
        # (the align that just finished registered cloud nxt as the source)
        return {"value": round(steps / el, 3), "ms_per_step": round(el / steps * 1e3, 5), "max_abs_pose_difference_to_the_sequential_loop": float(np.abs(last[0] - T_seq[nxt[0]]).max())}

    dev = loop(False, steps, warmup)
    host = loop(True, max(5, steps // 4), 2)
    out = dict(dev)
    if not args.no_pipelined_leg:
        try:
            out["pipelined"] = pipelined_loop(steps, warmup)
        except Exception as ex:  # noqa: BLE001
            out["pipelined"] = {"error": repr(ex)}
    out.update({"metric": "registrations/sec (100-iter reuse) through the reference-API classes, default neighbour enum", "unit": "registrations/sec", "n_gpus": 1, "steps": steps, "warmup": warmup,
                "repeats": REPEATS, "higher_is_better": True, "dtype": "f64", "data": "bundled scans (real LiDAR)",
                "config": {"workload": desc, "method": "VGICP (pygicp.FastVGICPCuda, defaults)", "neighbor_search": "DIRECT27", "covariance": "CPU_PARALLEL_KDTREE enum, served by the device's exact k-NN",
                           "loop": "100times_reuse (align.cpp:87-101), host clouds in, numpy -> PointCloud conversion included"},
                "host_kdtree_opt_in": host["value"], "host_kdtree": host})
    return out


EXTRA_CONFIGS = {
    # name: (workload, cov, search, steps, warmup, cpu seconds)
    # BASELINE configs[1] with the other two covariance modes of align.cpp:190-213: the headline is gpu_bruteforce
    "bundled17k_rbf_kernel": ("bundled17k", "rbf", "DIRECT27", 100, 10, 0.0),
    "bundled17k_parallel_kdtree": ("bundled17k", "kdtree", "DIRECT27", 20, 3, 0.0),
    "synth100k_rbf": ("synth100k", "rbf", "DIRECT27", 40, 5, 10.0),
    "synth1m": ("synth1m", "knn", "DIRECT7", 40, 5, 12.0),
}


def main():
    args = parse()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launch the driver would have made (one rank per GPU over RCCL)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the process group decides (one rank per GPU)" % (args.gpus, world), file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload == "lidar_stream":
        if world != 1:
            raise SystemExit("lidar_stream is a single-GPU workload")
        emit(run_stream(args, args.steps, args.warmup))
        return
    if args.workload == "fgicp17k":
        if world != 1:
            raise SystemExit("fgicp17k is a single-GPU workload")
        emit(run_fgicp(args, args.steps, args.warmup, local_rank))
        return
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

    if args.search is None:
        args.search = "DIRECT7" if args.workload == "synth1m" else "DIRECT27"
    sharded, sharded_hung = None, False
    if world > 1:
        # N > 1: `value` is the path north_star describes -- ONE registration sharded over the GPUs (BASELINE configs[4]), strong scaling. The N
        # independent 17k streams the line used to lead with (no traffic between the GPUs: weak scaling by construction) follow as `replicas_17k`.
        # Every collective leg runs on a worker thread under a deadline: a stuck exchange must not take the JSON line with it.
        head, hung = run_with_deadline(lambda: sharded_headline(args, dist, rank, world, local_rank, dev, small=share_gpu), args.sharded_deadline)
        if hung or head is None:
            head = None
            sharded_hung = hung
        replicas = None
        if not sharded_hung:
            replicas = run_registration(args, args.workload, args.cov, args.search, min(args.steps, 100), min(args.warmup, 10), local_rank, dist, dev, world, rank, headline=False,
                                        cpu_budget=args.cpu_seconds, cpu=False)
            if os.environ.get("FVH_BENCH_SHARDED_DETAIL", "1") == "1" and (not share_gpu or os.environ.get("FVH_BENCH_SHARDED_TEST") == "1"):
                sharded, sharded_hung = run_with_deadline(lambda: sharded_leg(args, dist, rank, world, local_rank, dev, small=share_gpu), args.sharded_deadline)
                if sharded_hung:
                    sharded = {"error": "sharded detail leg did not finish within %d s on rank %d" % (args.sharded_deadline, rank)}
        if rank != 0:
            finish(dist, sharded_hung)
            return
        if head is not None:
            out = head
            if replicas is not None:
                out["replicas_17k"] = {k: replicas[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling", "config", "fitness_score", "per_registration", "roofline") if k in replicas}
                out["replicas_17k"]["note"] = "N independent registration streams of the 17k pair, one per GPU (the N = 1 headline replicated): no data-path collective, weak scaling by construction"
        else:  # the sharded step failed or hung: say so, and report what was measured
            out = replicas if replicas is not None else {"metric": "registrations/sec", "value": None, "unit": "registrations/sec", "n_gpus": world}
            out["sharded_headline_error"] = "the sharded registration did not finish within %d s: `value` is N independent 17k streams instead" % args.sharded_deadline
        if sharded is not None:
            out["sharded"] = sharded
    else:
        out = run_registration(args, args.workload, args.cov, args.search, args.steps, args.warmup, local_rank, dist, dev, world, rank, headline=True, cpu_budget=args.cpu_seconds)

    # ---- the other single-GPU configurations of BASELINE.json, time-boxed ----
    default_headline = args.workload == "bundled17k" and args.cov == "knn" and world == 1
    names = [] if args.configs == "none" else (args.configs.split(",") if args.configs else (["bundled17k_rbf_kernel", "bundled17k_parallel_kdtree", "bundled17k_fgicp", "synth100k_rbf", "synth1m", "lidar_stream"] if default_headline else []))
    if names:
        configs = {}
        for name in names:
            if time.perf_counter() - T_START > args.time_box:
                configs[name] = {"skipped": "time box of %.0f s reached before this configuration started" % args.time_box}
                continue
            try:
                if name == "lidar_stream":
                    configs[name] = run_stream(args, 60, 5)
                elif name == "bundled17k_parallel_kdtree":
                    configs[name] = run_reference_api(args, 60, 5)
                elif name == "bundled17k_fgicp":
                    configs[name] = run_fgicp(args, 60, 5, local_rank)
                else:
                    wl, cov, search, steps, warmup, cpu_s = EXTRA_CONFIGS[name]
                    configs[name] = run_registration(args, wl, cov, search, steps, warmup, local_rank, cpu_budget=cpu_s, cpu=cpu_s > 0)
            except Exception as ex:  # the headline must not depend on an extra configuration
                configs[name] = {"error": repr(ex)}
        if default_headline and not args.no_cpu_baseline and time.perf_counter() - T_START <= args.time_box + 30:
            try:
                configs["cpu_fastvgicp_direct1"] = cpu_config0()
            except Exception as ex:
                configs["cpu_fastvgicp_direct1"] = {"error": repr(ex)}
        out["configs"] = configs
    out["bench_wall_s"] = round(time.perf_counter() - T_START, 1)
    emit(out)
    finish(dist, sharded_hung)


if __name__ == "__main__":
    main()
