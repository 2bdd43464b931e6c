#!/bin/bash
mkdir -p gpurun_out/r03s
python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^RCCL|^HIP version|^ROCm|^Hostname|^Librccl|amdgpu.ids" | tail -6 > gpurun_out/r03s/pytest_full.txt
cat gpurun_out/r03s/pytest_full.txt
FVH_COMMIT=05f4b179e1e1 bash tools/r03_artifacts.sh all
