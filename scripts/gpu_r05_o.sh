#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_o; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|create +'
timeout 120 python scripts/time_c3.py 10000 0 2>&1 | grep -v "$F" | tee -a "$OUT/c3_merged.txt"
RXHIP_TEST_HOOKS=1 RXHIP_SWEEP_MERGED=0 timeout 120 python scripts/time_c3.py 10000 0 2>&1 | grep -v "$F" | sed 's/^lib default/two launches/' | tee -a "$OUT/c3_merged.txt"
for seg in 1250 768 512; do timeout 120 python scripts/time_c3.py 10000 $seg 2>&1 | grep -v "$F" | tee -a "$OUT/c3_merged.txt"; done
timeout 900 python -m pytest tests/test_seeded_tile_inverse_gpu.py tests/test_fixed_point_adversarial_gpu.py tests/test_node_marginals.py tests/test_dense_missing_parallel_gpu.py tests/test_badly_scaled_models_gpu.py tests/test_random_shapes_gpu.py -m gpu -q 2>&1 | grep -v "$F" | tail -8 | tee "$OUT/pytest.txt"
