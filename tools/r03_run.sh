set -x
O=gpurun_out/r03n
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python tools/ab_bench.py --workload synth1m --steps 50 default:FVH_COST_CHUNKED_ITEMS=0 default default:FVH_COST_CHUNKED_ITEMS=0 default > $O/ab1m.txt 2>&1
tail -6 $O/pytest.txt; cat $O/ab1m.txt
