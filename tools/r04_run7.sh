#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_peer.py tests/test_gpu_parity.py tests/test_gpu_side_stream.py tests/test_gpu_pygicp.py -m gpu -q 2>&1 | tail -12 > $O/tests.txt
timeout 300 python tools/ab_bench.py --workload synth100k --cov rbf --steps 40 default default:FVH_COST_PRIO=6 default:FVH_COST_PRIO=0 > $O/abprio_100k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload synth1m --steps 40 default default:FVH_COST_PRIO=6 > $O/abprio_1m.txt 2>&1
timeout 300 python tools/ab_bench.py --workload bundled17k --steps 200 default default:FVH_COST_PRIO=6 default:FVH_COST_PRIO=0 > $O/abprio_17k.txt 2>&1
tail -4 $O/tests.txt; cat $O/abprio_100k.txt $O/abprio_1m.txt $O/abprio_17k.txt
