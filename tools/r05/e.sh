O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cuda_compat.py -x -q -s 2>&1 | grep -v "^$" | tail -25
V=fast_gicp_amd/lib/variants
FVH_LIB_PATH=$V/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/persist_ndt.txt 2>&1
FVH_LIB_PATH=$V/timing/libfast_vgicp_hip.so timeout 120 python tools/main_timing.py --ndt > $O/main_ndt.txt 2>&1
FVH_LIB_PATH=$V/timing_lm/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/persist_ndt_lmstages.txt 2>&1
FVH_LIB_PATH=$V/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py > $O/persist_17k.txt 2>&1
FVH_LIB_PATH=$V/timing/libfast_vgicp_hip.so timeout 120 python tools/main_timing.py > $O/main_17k.txt 2>&1
head -30 $O/persist_ndt.txt; head -24 $O/main_ndt.txt; head -12 $O/persist_ndt_lmstages.txt
