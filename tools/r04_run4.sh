#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python tools/r04_conc.py "" FVH_SHARE_BY_XCD=0 FVH_CONFINED_SLOT_PCT=100 FVH_CONFINED_SLOT_PCT=34 FVH_XCD_LOCAL=0 > $O/conc.txt 2>&1
timeout 400 python tools/ab_bench.py --workload bundled17k --steps 150 default default:FVH_COST_PRIO=1 default:FVH_COST_PRIO=3 default:FVH_COST_PRIO=4 default:FVH_COST_PRIO=1 default > $O/abprio.txt 2>&1
timeout 300 python tools/ab_bench.py --workload lidar_stream --steps 60 default default:FVH_COST_PRIO=4 > $O/abprio_lidar.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth1m --steps 30 default default:FVH_COST_PRIO=4 default:FVH_COST_PRIO=1 > $O/abprio_1m.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth100k --cov rbf --steps 30 default default:FVH_COST_PRIO=4 default:FVH_COST_PRIO=1 > $O/abprio_100k.txt 2>&1
for P in 0 1 4; do FVH_COST_PRIO=$P FVH_LIB_PATH=$PWD/fast_gicp_amd/lib/variants/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py > $O/pt17k_prio$P.txt 2>&1; done
GICP_ALIGN_BREAKDOWN=1 timeout 120 ./fast_gicp_amd/apps/gicp_align data/251370668.pcd data/251371071.pcd > $O/gicp_align_breakdown.txt 2>&1
cat $O/conc.txt $O/abprio.txt $O/abprio_lidar.txt $O/abprio_1m.txt $O/abprio_100k.txt; tail -9 $O/pt17k_prio1.txt; grep -A1 "DIRECT27\|bruteforce) ---" $O/gicp_align_breakdown.txt | tail -8
