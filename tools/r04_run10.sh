#!/bin/bash
# round 4, GPU call 10: LM step on every workgroup at the large grids (3 workgroups per CU)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04q; mkdir -p $O
timeout 400 python tools/ab_bench.py --workload synth100k --cov rbf --steps 40 default default:FVH_LM_EVERYWHERE=2 ev0:FVH_LM_EVERYWHERE=2 default default:FVH_LM_EVERYWHERE=2 ev0:FVH_LM_EVERYWHERE=2 > $O/ab_100k.txt 2>&1
timeout 400 python tools/ab_bench.py --workload synth1m --steps 40 default default:FVH_LM_EVERYWHERE=2 ev0:FVH_LM_EVERYWHERE=2 default default:FVH_LM_EVERYWHERE=2 ev0:FVH_LM_EVERYWHERE=2 > $O/ab_1m.txt 2>&1
cat $O/ab_100k.txt $O/ab_1m.txt
