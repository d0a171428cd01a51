#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_p; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|create +'
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -12 | tee "$OUT/pytest_gpu.txt"
