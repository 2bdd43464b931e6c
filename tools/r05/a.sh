O=gpurun_out/r05a; mkdir -p $O
timeout 60 tools/probes/probe_valu_rate > $O/valu_rate.txt 2>&1; cat $O/valu_rate.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"
wc -c $O/bench_stdout.txt; wc -l $O/bench_stdout.txt; cat $O/bench_stdout.txt
cp bench_detail.json $O/bench_detail.json
timeout 300 python -m pytest tests/test_gpu_bench_flow.py -x -q 2>&1 | tail -5
