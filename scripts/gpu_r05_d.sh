#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_d; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
python scripts/time_c3.py 2>&1 | grep -v "$F" | tee "$OUT/time_c3.txt"
timeout 900 python -m pytest tests/test_tree_engine_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -60 | tee "$OUT/pytest_tree.txt"
