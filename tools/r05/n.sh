O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ndt.py tests/test_gpu_streaming.py tests/test_gpu_xcd_local.py tests/test_gpu_parity_large.py tests/test_gpu_correspondences.py tests/test_gpu_cuda_compat.py -x -q 2>&1 | tail -8
for s in 1 0 1 0; do FVH_COST_SPLIT=$s timeout 200 python bench.py --workload lidar_stream --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
c=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$s', c['value'], c['roofline']['avg_launch_us'], c.get('configs'))"; python -c "
import json; d=json.load(open('bench_detail.json')); print('   pipelined', d['pipelined']['registrations_per_sec'])"; done
