O=gpurun_out/r05b; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_streaming.py -x -q 2>&1 | tail -15
timeout 200 python bench.py --workload lidar_stream --steps 60 --warmup 5 --no-cpu-baseline > $O/stream_stdout.txt 2> $O/stream_stderr.txt; echo "stream rc=$?"; cat $O/stream_stdout.txt
cp bench_detail.json $O/stream_detail.json
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
