#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_j; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|create +'
for seg in 0 768 640 512 400 333 256; do
  python scripts/time_c3.py 10000 $seg 2>&1 | grep -v "$F" | tee -a "$OUT/c3_segments.txt"
done
timeout 900 python -m pytest tests/test_seeded_tile_inverse_gpu.py tests/test_fixed_point_adversarial_gpu.py tests/test_node_marginals.py tests/test_dense_missing_parallel_gpu.py tests/test_lgssm_gpu.py tests/test_headline_parity_gpu.py -m gpu -q 2>&1 | grep -v "$F" | tail -8 | tee "$OUT/pytest.txt"
