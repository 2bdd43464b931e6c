"""bench.py's N > 1 control flow (barrier, max over ranks, one JSON line from rank 0) walked on a single-GPU box: both ranks on
device 0, gloo for the barrier (FVH_BENCH_SHARE_GPU / FVH_BENCH_BACKEND, test-only knobs). With N > 1 the line's `value` is ONE registration
stream sharded over the ranks (BASELINE configs[4] on the driver's node; here its small form on the peer route -- RCCL needs a GPU per rank),
strong scaling, with the same step on one GPU next to it; the N independent 17k streams follow as `replicas_17k`, the per-route detail as
`sharded`. Real xGMI scaling is the driver's to measure."""
import json
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu


def test_two_rank_bench_prints_one_aggregate_line():
    env = dict(os.environ, FVH_BENCH_SHARE_GPU="1", FVH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", FVH_BENCH_SHARDED_TEST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--configs", "none"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=util.ROOT)
    lines = [l for l in p.stdout.splitlines() if l.strip()]  # (gloo prints its own "[Gloo] Rank 0 is connected ..." lines to stdout)
    assert p.returncode == 0 and lines and lines[-1].startswith("{") and sum(l.startswith("{") for l in lines) == 1, (p.returncode, p.stdout[-400:], p.stderr[-800:])
    assert len(lines[-1]) < 4096  # the line the driver parses: the LAST stdout line, compact
    c = json.loads(lines[-1])
    assert c["n_gpus"] == 2 and c["steps"] == 10 and c["warmup"] == 2 and c["scaling"] == "strong" and c["higher_is_better"] is True
    assert c["value"] > 0 and abs(c["value"] - 1e3 / c["ms_per_step"]) <= 1e-3 * c["value"]  # ONE stream: steps / max-over-ranks time (median repeat)
    assert c["repeats"] >= 5 and c["ms_per_step_min"] <= c["ms_per_step"] <= c["ms_per_step_max"]
    assert c["config"]["route"] == "peer" and "sharded by spatial tile" in c["config"]["parallelism"]
    assert c["roofline"]["bound"] == "hbm" and c["roofline"]["frac"] > 0 and c["roofline"]["avg_launch_us"] > 0 and c["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert c["cpu_baseline"] is None and c["n1_same_workload"]["value"] > 0 and c["replicas_17k"]["value"] > 0
    # the full detail: bench_detail.json beside the script (and the "bench_detail: " line on stderr)
    det = [l for l in p.stderr.splitlines() if l.startswith("bench_detail: ")]
    assert len(det) == 1
    d = json.loads(det[0][len("bench_detail: "):])
    assert d == json.load(open(os.path.join(util.ROOT, "bench_detail.json")))
    assert d["value"] == c["value"]
    assert d["n1_same_workload"]["value"] > 0 and d["n1_same_workload"]["converged"]
    assert d["per_registration"]["converged"] and d["per_registration"]["pose_equals_single_gpu"], d["per_registration"]
    # (both are rounded to three decimals; two processes sharing ONE GPU can be arbitrarily slow: absolute + relative tolerance)
    assert abs(d["speedup_vs_one_gpu"] - d["value"] / d["n1_same_workload"]["value"]) <= 1e-3 + 2e-3 * d["speedup_vs_one_gpu"]
    rep = d["replicas_17k"]  # the N = 1 headline, replicated: whole-job aggregate = ranks x steps / max time
    assert rep["scaling"] == "weak" and rep["value"] > 0 and abs(rep["value"] - 2 * 1e3 / rep["ms_per_step"]) <= 1e-3 * rep["value"]
    assert abs(rep["fitness_score"] - 0.198792) < 1e-5
    assert "roofline" in rep and rep["roofline"]["bound"] in ("hbm", "mfma")
    sh = d["sharded"]["small"]
    assert "error" not in sh and "error" not in sh["peer"], sh
    assert sh["single_gpu"]["converged"] and sh["peer"]["converged"] and sh["peer"]["pose_equals_single_gpu"], sh
    assert sh["peer"]["ms_per_registration"] > 0 and sh["peer"]["source_stage_ms"] > 0 and sh["peer"]["align_ms"] > 0
    assert "rccl" not in sh  # (one GPU: not runnable here)
