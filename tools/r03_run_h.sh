set -x
O=gpurun_out/r03h
mkdir -p $O
timeout 120 python tools/probe_determinism.py > $O/determinism.txt 2>&1
FVH_LIB_PATH=fast_gicp_amd/lib/variants/r02/libfast_vgicp_hip.so timeout 120 python tools/probe_determinism.py > $O/determinism_r02.txt 2>&1
FVH_LIB_PATH=fast_gicp_amd/lib/variants/r02timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing_r02.py > $O/pt17k_r02.txt 2>&1
cat $O/determinism.txt $O/determinism_r02.txt $O/pt17k_r02.txt
