#!/bin/bash
# round 4, GPU call 11: small grids, ONE level, every workgroup adds all rows and runs the LM step (one hand-off per trip)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04q; mkdir -p $O
timeout 400 python tools/ab_bench.py --workload lidar_stream --steps 120 default default:FVH_SMALL_GRID_LAYOUT=0,FVH_LM_EVERYWHERE_SINGLE=1 default:FVH_SMALL_GRID_LAYOUT=0 default default:FVH_SMALL_GRID_LAYOUT=0,FVH_LM_EVERYWHERE_SINGLE=1 > $O/ab_lidar3.txt 2>&1
FVH_SMALL_GRID_LAYOUT=0 FVH_LM_EVERYWHERE_SINGLE=1 timeout 600 python -m pytest tests/test_gpu_ndt.py tests/test_gpu_streaming.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" > $O/tests3.txt
cat $O/ab_lidar3.txt $O/tests3.txt
