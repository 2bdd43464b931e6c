export FVH_COMMIT=ebfca5045832
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
bash tools/r05_artifacts.sh pmc 2>&1 | tail -16
bash tools/r05_artifacts.sh stats 2>&1 | tail -3
O=gpurun_out/r05m; mkdir -p $O
V=fast_gicp_amd/lib/variants
FVH_LIB_PATH=$V/timing/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/persist_timing_ndt.txt 2>&1
FVH_LIB_PATH=$V/timing_lm/libfast_vgicp_hip.so timeout 120 python tools/persist_timing.py --ndt > $O/persist_timing_ndt_lmstages.txt 2>&1
head -12 $O/persist_timing_ndt.txt
