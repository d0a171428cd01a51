#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_c; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
python scripts/time_c3.py 2>&1 | grep -v "$F" | tee "$OUT/time_c3.txt"
RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_r4base.so python scripts/time_c3.py 2>&1 | grep -v "$F" | tee -a "$OUT/time_c3.txt"
timeout 900 python -m pytest tests/test_fixed_point_adversarial_gpu.py tests/test_seeded_tile_inverse_gpu.py tests/test_device_tables_gpu.py tests/test_badly_scaled_models_gpu.py -m gpu -q 2>&1 | grep -v "$F" | tail -15 | tee "$OUT/pytest.txt"
