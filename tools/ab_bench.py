#!/usr/bin/env python
"""A/B timing of library variants (tools/build_variants.py) and launch-shape knobs on the GPU box: runs bench.py once per
(variant, env) combination and prints one line each: registrations/s, cost-kernel average launch, k-NN, voxel map.
    python tools/ab_bench.py [--workload W] [--steps N] variant[:ENV=VAL,ENV=VAL] ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    workload, steps, streams, cov = "bundled17k", "100", "1", "knn"
    while args and args[0].startswith("--"):
        k = args.pop(0)
        if k == "--workload":
            workload = args.pop(0)
        elif k == "--steps":
            steps = args.pop(0)
        elif k == "--cov":
            cov = args.pop(0)
        elif k == "--streams":  # > 1: also the concurrent-handles leg (S engine handles, S host threads)
            streams = args.pop(0)
    for spec in args:
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        if name != "default":
            env["FVH_LIB_PATH"] = os.path.join(ROOT, "fast_gicp_amd", "lib", "variants", name, "libfast_vgicp_hip.so")
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", steps, "--warmup", "10", "--no-cpu-baseline", "--streams", streams, "--configs", "none", "--no-host-leg", "--cov", cov]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(spec, "FAILED", p.stderr[-400:].replace("\n", " | "), flush=True)
            continue
        d = json.loads(line[-1])
        st = d.get("stages", {})
        nan = float("nan")
        cs = d.get("concurrent_streams") or {}
        print("%-52s %9.1f reg/s  cost %7.1f us (roofline launch %7.1f)  sort %5.1f  knn %6.1f  cov %5.1f  vm %5.1f  fitness %.6f aborts %s%s" % (
            spec, d["value"], st.get("cost", {}).get("avg_us", nan), (d.get("roofline") or {}).get("avg_launch_us", nan), st.get("sort", {}).get("avg_us", nan), st.get("knn", {}).get("avg_us", nan),
            st.get("cov", {}).get("avg_us", nan), st.get("voxelmap", {}).get("avg_us", nan), d.get("fitness_score", nan), (d.get("per_registration") or {}).get("persistent_launches_aborted_by_watchdog"),
            ("  | %s streams: %.1f reg/s" % (cs.get("streams"), cs.get("registrations_per_sec", nan))) if cs else ""), flush=True)


if __name__ == "__main__":
    main()
