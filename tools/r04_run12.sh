#!/bin/bash
# round 4, GPU call 12: two-launch radix passes for mid-size clouds (FVH_SORT_FUSED_BITS: 0 = four-launch passes, 9, 10)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spatial_order.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.txt
FVH_SORT_FUSED_BITS=10 timeout 600 python -m pytest tests/test_gpu_spatial_order.py -m gpu -q -x 2>&1 | tail -5 >> $O/tests.txt
cat $O/tests.txt
timeout 500 python tools/ab_bench.py --workload synth100k --cov rbf --steps 40 default:FVH_SORT_FUSED_BITS=0 default default:FVH_SORT_FUSED_BITS=10 default:FVH_SORT_FUSED_BITS=0 default default:FVH_SORT_FUSED_BITS=10 > $O/ab_100k.txt 2>&1
timeout 500 python tools/ab_bench.py --workload synth1m --steps 40 default:FVH_SORT_FUSED_BITS=0 default default:FVH_SORT_FUSED_BITS=10 default:FVH_SORT_FUSED_BITS=0 default default:FVH_SORT_FUSED_BITS=10 > $O/ab_1m.txt 2>&1
cat $O/ab_100k.txt $O/ab_1m.txt
