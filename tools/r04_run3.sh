#!/bin/bash
# round 4, GPU call 3: concurrency (aligns share the chip XCD by XCD), the priority experiment, the gicp_align table
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_xcd_local.py tests/test_gpu_robustness.py tests/test_gpu_pygicp.py -m gpu -q 2>&1 | tail -30 > $O/tests.txt
timeout 400 python tools/ab_bench.py --workload bundled17k --steps 100 --streams 4 default default:FVH_SHARE_BY_XCD=0 default:FVH_CONFINED_SLOT_PCT=50 default:FVH_CONFINED_SLOT_PCT=100 default:FVH_COST_PRIO=1 default:FVH_COST_PRIO=2 > $O/ab17k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload bundled17k --steps 100 --streams 8 default default:FVH_SHARE_BY_XCD=0 default:FVH_SLOT_MAX_SPLIT=8 > $O/ab17k_s8.txt 2>&1
timeout 200 python tools/ab_bench.py --workload bundled17k --steps 100 --streams 2 default default:FVH_SHARE_BY_XCD=0 > $O/ab17k_s2.txt 2>&1
for i in 1 2; do timeout 120 ./fast_gicp_amd/apps/gicp_align data/251370668.pcd data/251371071.pcd > $O/gicp_align_$i.txt 2>&1; done
tail -4 $O/tests.txt; cat $O/ab17k.txt $O/ab17k_s8.txt $O/ab17k_s2.txt; cat $O/gicp_align_2.txt
