set -x
O=gpurun_out/r03m
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python tools/ab_bench.py --workload synth1m --steps 50 default:FVH_BITMAP_MIN_POINTS=100000000 default default:FVH_BITMAP_MIN_POINTS=100000000 default > $O/ab1m.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth100k --steps 50 default default:FVH_BITMAP_MIN_POINTS=1 default default:FVH_BITMAP_MIN_POINTS=1 > $O/ab100k.txt 2>&1
timeout 300 python tools/ab_bench.py --steps 200 default default:FVH_BITMAP_MIN_POINTS=1 default default:FVH_BITMAP_MIN_POINTS=1 > $O/ab17k.txt 2>&1
tail -6 $O/pytest.txt; cat $O/ab1m.txt $O/ab100k.txt $O/ab17k.txt
