#!/bin/bash
# round 4, GPU call 8: the LM step on every workgroup (FVH_LM_EVERYWHERE) -- NDT stream and the 17k headline; poll spacing of the non-collectors
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04q; mkdir -p $O
timeout 400 python tools/ab_bench.py --workload lidar_stream --steps 120 default default:FVH_LM_EVERYWHERE=1 ev0:FVH_LM_EVERYWHERE=1 ev1:FVH_LM_EVERYWHERE=1 ev8:FVH_LM_EVERYWHERE=1 default:FVH_LM_EVERYWHERE=1 ev0:FVH_LM_EVERYWHERE=1 > $O/ab_lidar2.txt 2>&1
timeout 400 python tools/ab_bench.py --workload bundled17k --steps 200 default default:FVH_LM_EVERYWHERE=2 ev0:FVH_LM_EVERYWHERE=2 ev1:FVH_LM_EVERYWHERE=2 ev8:FVH_LM_EVERYWHERE=2 default:FVH_LM_EVERYWHERE=2 ev0:FVH_LM_EVERYWHERE=2 > $O/ab_17k2.txt 2>&1
FVH_LM_EVERYWHERE=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" > $O/tests2.txt
cat $O/ab_lidar2.txt $O/ab_17k2.txt $O/tests2.txt
