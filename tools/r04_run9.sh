#!/bin/bash
# round 4, GPU call 9: per-trip timeline with the LM step on every workgroup
O=gpurun_out/r04q; mkdir -p $O
V=$PWD/fast_gicp_amd/lib/variants/timing/libfast_vgicp_hip.so
FVH_LIB_PATH=$V timeout 120 python tools/persist_timing.py > $O/pt17k_base.txt 2>&1
FVH_LM_EVERYWHERE=2 FVH_LIB_PATH=$V timeout 120 python tools/persist_timing.py > $O/pt17k_everywhere.txt 2>&1
FVH_LIB_PATH=$V timeout 120 python tools/persist_timing.py --ndt > $O/ptndt_base.txt 2>&1
FVH_LM_EVERYWHERE=1 FVH_LIB_PATH=$V timeout 120 python tools/persist_timing.py --ndt > $O/ptndt_everywhere.txt 2>&1
head -12 $O/pt17k_base.txt; head -12 $O/pt17k_everywhere.txt
