set -x
O=gpurun_out/r03c
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
for v in timing timing_twice timing_backoff; do
  FVH_LIB_PATH=fast_gicp_amd/lib/variants/$v/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py > $O/pt17k_$v.txt 2>&1
  FVH_LIB_PATH=fast_gicp_amd/lib/variants/$v/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/ptndt_$v.txt 2>&1
done
timeout 400 python tools/ab_bench.py --steps 200 default backoff60 backoff100 default backoff60 backoff100 > $O/ab17k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload lidar_stream --steps 100 default backoff60 backoff100 default backoff60 > $O/ab_stream.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth100k --steps 50 default backoff100 > $O/ab100k.txt 2>&1
tail -5 $O/pytest.txt; cat $O/ab17k.txt $O/ab_stream.txt $O/ab100k.txt
