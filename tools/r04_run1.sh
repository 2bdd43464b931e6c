#!/bin/bash
# round 4, GPU call 1: the whole GPU suite on the replicated-collector LM kernel + A/B of the hand-off flavours and grid layouts
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04b; mkdir -p $O

timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -40 > $O/tests.txt
timeout 300 python tools/ab_bench.py --workload bundled17k --steps 100 --streams 4 default default:FVH_XCD_LOCAL=0 default:FVH_SHARE_BY_XCD=0 > $O/ab17k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload lidar_stream --steps 60 default:FVH_SMALL_GRID_LAYOUT=0 default:FVH_SMALL_GRID_LAYOUT=1 default:FVH_SMALL_GRID_LAYOUT=2 default:FVH_SMALL_GRID_LAYOUT=2,FVH_XCD_LOCAL=0 > $O/ablidar.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth1m --steps 30 default default:FVH_XCD_LOCAL=0 > $O/ab1m.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth100k --cov rbf --steps 30 default default:FVH_XCD_LOCAL=0 > $O/ab100k.txt 2>&1
FVH_LIB_PATH=$PWD/fast_gicp_amd/lib/variants/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py > $O/pt17k.txt 2>&1
FVH_LIB_PATH=$PWD/fast_gicp_amd/lib/variants/timing/libfast_vgicp_hip.so FVH_XCD_LOCAL=0 timeout 120 python tools/persist_timing.py > $O/pt17k_nolocal.txt 2>&1
for L in 0 1 2; do FVH_SMALL_GRID_LAYOUT=$L FVH_LIB_PATH=$PWD/fast_gicp_amd/lib/variants/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/ptndt_layout$L.txt 2>&1; done
tail -5 $O/tests.txt; cat $O/ab17k.txt $O/ablidar.txt $O/ab1m.txt $O/ab100k.txt; tail -12 $O/pt17k.txt
