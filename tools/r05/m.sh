O=gpurun_out/r05m; mkdir -p $O
V=fast_gicp_amd/lib/variants
FVH_LIB_PATH=$V/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/persist_timing_ndt.txt 2>&1
FVH_LIB_PATH=$V/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py > $O/persist_timing_17k.txt 2>&1
FVH_LIB_PATH=$V/timing_lm/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/persist_timing_ndt_lmstages.txt 2>&1
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cp bench_detail.json $O/bench_detail.json; cat $O/bench_line.json | cut -c1-1500
export FVH_BENCH_SHARE_GPU=1 FVH_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 FVH_BENCH_SHARDED_TEST=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --configs none > $O/two_ranks_line.txt 2> $O/two_ranks.err; echo "two ranks rc=$?"; cp bench_detail.json $O/two_ranks_detail.json; tail -1 $O/two_ranks_line.txt | cut -c1-600
head -14 $O/persist_timing_ndt.txt
