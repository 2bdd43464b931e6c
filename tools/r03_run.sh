#!/bin/bash
for i in 1 2; do
python tools/ab_bench.py --steps 300 default default:FVH_COST_TARGET_ITEMS=70000 default:FVH_COST_TARGET_ITEMS=52000 default:FVH_COST_TARGET_ITEMS=90000 default:FVH_COST_TARGET_ITEMS=200000
done
