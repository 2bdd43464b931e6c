set -x
O=gpurun_out/r03g
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
timeout 500 python tools/ab_bench.py --steps 200 r02 default rep8 rep16 rep64 default rep8 rep16 rep64 > $O/ab17k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload lidar_stream --steps 100 default rep8 rep16 default rep8 rep16 > $O/ab_stream.txt 2>&1
tail -5 $O/pytest.txt; cat $O/ab17k.txt $O/ab_stream.txt
