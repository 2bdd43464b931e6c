O=gpurun_out/r05m; mkdir -p $O
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cp bench_detail.json $O/bench_detail.json; cat $O/bench_line.json | cut -c1-3200
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -3
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
