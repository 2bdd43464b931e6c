set -x
O=gpurun_out/r03o
mkdir -p $O
timeout 300 python tools/ab_bench.py --workload synth1m --steps 50 default wg4 default wg4 > $O/ab1m.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth100k --steps 50 default wg4 > $O/ab100k.txt 2>&1
timeout 300 python tools/ab_bench.py --steps 100 default wg4 > $O/ab17k.txt 2>&1
cat $O/ab1m.txt $O/ab100k.txt $O/ab17k.txt
