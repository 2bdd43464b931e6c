#!/bin/bash
# round 4, GPU call 16: strided dispatch order of the 17k k-NN (FVH_KNN_PERMUTE)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04s; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "knn or neighbor or cooperative" 2>&1 | grep -E "passed|failed" > $O/tests.txt; cat $O/tests.txt
timeout 500 python tools/ab_bench.py --workload bundled17k --steps 200 default:FVH_KNN_PERMUTE=0 default default:FVH_KNN_PERMUTE=0 default > $O/ab_17k_perm.txt 2>&1
cat $O/ab_17k_perm.txt
FVH_LIB_PATH=$PWD/fast_gicp_amd/lib/variants/knntiming/libfast_vgicp_hip.so timeout 200 python tools/knn_timing.py 2>&1 | grep -E "kernel span|wave starts|duration perc" 
FVH_LIB_PATH=$PWD/fast_gicp_amd/lib/variants/knntiming/libfast_vgicp_hip.so timeout 200 python tools/knn_timing.py target 2>&1 | grep -E "kernel span|wave starts|duration perc" 
FVH_KNN_PERMUTE=0 FVH_LIB_PATH=$PWD/fast_gicp_amd/lib/variants/knntiming/libfast_vgicp_hip.so timeout 200 python tools/knn_timing.py target 2>&1 | grep -E "kernel span|wave starts|duration perc" 
