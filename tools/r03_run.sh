set -x
O=gpurun_out/r03j
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -30 $O/pytest.txt; tail -3 $O/bench.err; python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("headline", d["value"], d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "wall", d.get("bench_wall_s"))
for k, v in (d.get("configs") or {}).items():
    if "value" in v:
        print(k, v["value"], v.get("ms_per_step"), "roofline", (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("avg_launch_us"), "cpu", (v.get("cpu_baseline") or {}).get("value"), (v.get("cpu_baseline") or {}).get("cores"), v.get("this_box_at_16_threads"))
    else:
        print(k, v)
PY
