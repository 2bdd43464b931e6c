#!/bin/bash
mkdir -p gpurun_out/r03s
python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^RCCL|^HIP version|^ROCm|^Hostname|^Librccl|amdgpu.ids" | tail -8 > gpurun_out/r03s/pytest_full.txt
cat gpurun_out/r03s/pytest_full.txt
python tools/ab_bench.py --steps 300 default default
python tools/ab_bench.py --workload lidar_stream --steps 100 default
