#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -40 > $O/tests.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -5 $O/tests.txt; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04f/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps")}, d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("concurrent_streams"))
for k, v in d["configs"].items():
    print(k, v.get("value"), (v.get("roofline") or {}).get("avg_launch_us"), v.get("error"), v.get("skipped"))
print(d["bench_wall_s"])
PY
tail -3 $O/bench.err
