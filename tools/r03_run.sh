set -x
O=gpurun_out/r03l
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_cuda_compat.py tests/test_gpu_robustness.py -m gpu -q 2>&1 | tail -15 > $O/pytest.txt
timeout 200 python bench.py --precision fp32 --configs none --no-cpu-baseline --streams 1 > $O/bench_fp32.json 2> $O/bench_fp32.err
timeout 200 python bench.py --configs none --no-cpu-baseline --streams 1 > $O/bench_fp64.json 2> $O/bench_fp64.err
timeout 200 python bench.py --workload lidar_stream --precision fp32 --no-cpu-baseline > $O/stream_fp32.json 2> $O/stream_fp32.err
tail -8 $O/pytest.txt
python - <<PY
import json
for n in ("bench_fp32", "bench_fp64", "stream_fp32"):
    d = json.load(open("$O/%s.json" % n))
    print(n, d["value"], d["dtype"], (d.get("roofline") or {}).get("avg_launch_us"), d.get("fitness_score"), (d.get("stages") or {}).get("cost"))
PY
