#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_g; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
timeout 600 python -m pytest tests/test_tree_engine_gpu.py -m gpu -q 2>&1 | grep -v "$F" | tail -5 | tee "$OUT/pytest.txt"


