set -x
O=gpurun_out/r03f
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
for v in timing; do
  FVH_LIB_PATH=fast_gicp_amd/lib/variants/$v/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py > $O/pt17k_$v.txt 2>&1
  FVH_LIB_PATH=fast_gicp_amd/lib/variants/$v/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/ptndt_$v.txt 2>&1
done
timeout 400 python tools/ab_bench.py --steps 200 r02 default r02 default > $O/ab17k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload lidar_stream --steps 100 r02 default r02 default > $O/ab_stream.txt 2>&1

tail -5 $O/pytest.txt; cat $O/ab17k.txt $O/ab_stream.txt
