O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"; cat $O/bench_stdout.txt
cp bench_detail.json $O/bench_detail.json
