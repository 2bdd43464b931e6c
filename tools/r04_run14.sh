#!/bin/bash
# round 4, GPU call 14: first look of the non-collectors at the group rows delayed by s_sleep N (x 64 cycles)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04q; mkdir -p $O
timeout 500 python tools/ab_bench.py --workload bundled17k --steps 200 default fs16 fs32 fs48 fs64 default fs16 fs32 fs48 fs64 > $O/ab_17k_fs.txt 2>&1
timeout 500 python tools/ab_bench.py --workload lidar_stream --steps 120 default fs16 fs32 fs48 fs64 default fs32 > $O/ab_lidar_fs.txt 2>&1
cat $O/ab_17k_fs.txt $O/ab_lidar_fs.txt
