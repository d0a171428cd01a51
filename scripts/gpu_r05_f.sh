#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_f; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
timeout 600 python -m pytest tests/test_tree_engine_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -5 | tee "$OUT/pytest_tree.txt"
for cfg in "256 4096 5 1" "256 4096 5 2" "128 32768 3 1" "128 32768 3 2" "128 131072 3 2" "64 262144 3 2"; do
  python scripts/prof_tree.py $cfg 2>&1 | grep -v "$F" | tee -a "$OUT/tree_modes.txt"
done
