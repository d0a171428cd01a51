#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_noise; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 600 python -m pytest tests/test_noise_vmp_gpu.py tests/test_lgssm_gpu.py tests/test_mvgmm_gpu.py -x -q 2>&1 | grep -v "$F" | tail -25 | tee "$OUT/pytest.txt"
