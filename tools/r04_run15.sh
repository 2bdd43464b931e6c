#!/bin/bash
# round 4, GPU call 15: points per wave of the two-launch radix passes (256 / 512 / 1024)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04r; mkdir -p $O
timeout 500 python tools/ab_bench.py --workload synth1m --steps 40 default si256 si1024 default si256 si1024 > $O/ab_1m_items.txt 2>&1
cat $O/ab_1m_items.txt
