set -x
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
FVH_LIB_PATH=fast_gicp_amd/lib/variants/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py > $O/persist_timing_17k.txt 2>&1
FVH_LIB_PATH=fast_gicp_amd/lib/variants/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/persist_timing_ndt.txt 2>&1
timeout 300 python tools/ab_bench.py --steps 200 r02 default r02 default > $O/ab17k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload lidar_stream --steps 100 r02 default r02 default > $O/ab_stream.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth100k --steps 50 r02 default > $O/ab100k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth1m --steps 50 r02 default > $O/ab1m.txt 2>&1
tail -5 $O/pytest.txt; cat $O/ab17k.txt $O/ab_stream.txt $O/ab100k.txt $O/ab1m.txt
