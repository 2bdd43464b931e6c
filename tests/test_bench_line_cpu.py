"""The line the driver parses (bench.py: compact_line): strict JSON, the contract's keys, shorter than 4 KB -- for the N = 1 and the N > 1 shapes
of the detail dictionary (tests/golden/bench_detail_n{1,2}.json are detail dictionaries bench.py produced on the GPU box), and for hostile ones."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports no torch at module level)

GOLD = os.path.join(ROOT, "tests", "golden")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")


def strict(line):
    def no_constants(c):
        raise ValueError("non-strict JSON constant %s" % c)
    return json.loads(line, parse_constant=no_constants)


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


@pytest.mark.parametrize("name", ["bench_detail_n1.json", "bench_detail_n2.json"])
def test_compact_line_carries_the_contract(name):
    detail = load(name)
    line = bench.compact_line(detail)
    assert len(line) < 4096 and "\n" not in line
    out = strict(line)
    for k in CONTRACT:
        assert k in out, k
    assert out["value"] == detail["value"] and out["ms_per_step"] == detail["ms_per_step"]
    assert out["steps"] == detail["steps"] and out["warmup"] == detail["warmup"] and out["n_gpus"] == detail["n_gpus"]
    assert "workload" in out["config"] and "model" not in out["config"]
    for k in ROOFLINE:
        assert k in out["roofline"], k
    assert out["roofline"]["frac"] == detail["roofline"]["frac"]
    assert out["roofline"]["avg_launch_us"] == detail["roofline"]["avg_launch_us"]
    if detail["n_gpus"] == 1:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in out["cpu_baseline"], k
        assert out["roofline"]["algorithmic_bytes_per_launch"] == detail["roofline"]["algorithmic_bytes_per_launch"]
        assert out["fitness_score"] == detail["fitness_score"]
        for cname, c in detail["configs"].items():  # one short entry per extra configuration
            assert out["configs"][cname]["value"] == c["value"]
            if isinstance(c.get("roofline"), dict):
                assert out["configs"][cname]["frac"] == c["roofline"]["frac"]
                assert out["configs"][cname]["avg_launch_us"] == c["roofline"].get("avg_launch_us")
        assert out["configs"]["lidar_stream"]["pipelined"] == detail["configs"]["lidar_stream"]["pipelined"]["registrations_per_sec"]
    else:
        assert out["cpu_baseline"] is None  # (N = 1 only)
        assert out["scaling"] == "strong" and out["n1_same_workload"]["value"] == detail["n1_same_workload"]["value"]


def test_compact_line_stays_short_whatever_the_detail_holds():
    d = load("bench_detail_n1.json")
    d["config"]["workload"] = "w" * 5000
    d["roofline"]["note"] = "n" * 100000
    d["roofline"]["kernel"] = "k" * 3000
    d["cpu_baseline"]["sample"] = "s" * 9000
    d["metric"] = "m" * 2000
    big = copy.deepcopy(d["configs"]["synth1m"])
    for i in range(200):
        d["configs"]["extra_%d" % i] = big
    d["configs"]["broken"] = {"error": "e" * 10000}
    d["ms_per_step_max"] = float("nan")
    d["roofline"]["traffic"] = float("inf")
    line = bench.compact_line(d)
    assert len(line) < 4096
    out = strict(line)
    for k in CONTRACT:
        assert k in out, k
    assert out["value"] == d["value"] and out["roofline"]["traffic"] is None


def test_compact_line_survives_missing_parts():
    out = strict(bench.compact_line({"metric": "registrations/sec", "value": None, "unit": "registrations/sec", "n_gpus": 2, "roofline": {"error": "x"}}))
    assert out["value"] is None and out["roofline"] == {"error": "x"} and out["cpu_baseline"] is None


def test_emit_prints_the_compact_line_last_and_alone_on_stdout(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "DETAIL_FILE", str(tmp_path / "bench_detail.json"))
    d = load("bench_detail_n1.json")
    bench.emit(d)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert strict(lines[0])["value"] == d["value"]
    assert cap.err.startswith("bench_detail: ")
    assert json.load(open(tmp_path / "bench_detail.json"))["stages"] == d["stages"]


def test_median_of():
    assert bench.median_of([3.0, 1.0, 2.0]) == (2.0, 1.0, 3.0)
    assert bench.median_of([4.0, 1.0, 2.0, 3.0]) == (2.5, 1.0, 4.0)
    assert bench.median_of([5.0]) == (5.0, 5.0, 5.0)
