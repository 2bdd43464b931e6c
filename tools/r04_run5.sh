#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -40 > $O/tests.txt
timeout 400 python tools/ab_bench.py --workload bundled17k --steps 200 --streams 4 default default:FVH_CONTENDED_SLOT_PCT=67 default:FVH_CONTENDED_SLOT_PCT=50 default:FVH_CONTENDED_SLOT_PCT=34 default > $O/ab17k.txt 2>&1
timeout 300 python tools/r04_conc.py "" FVH_CONTENDED_SLOT_PCT=67 FVH_CONTENDED_SLOT_PCT=50 > $O/conc.txt 2>&1
tail -5 $O/tests.txt; cat $O/ab17k.txt $O/conc.txt
