#!/bin/bash
# scratch: full GPU suite
mkdir -p gpurun_out/r03s
python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^RCCL|^HIP version|^ROCm|^Hostname|^Librccl|amdgpu.ids" | tail -15 > gpurun_out/r03s/pytest_full.txt
cat gpurun_out/r03s/pytest_full.txt
