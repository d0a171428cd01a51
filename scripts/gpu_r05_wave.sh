#!/bin/bash
# the executor above d = 8: parity tests, the register kernels' tests (unchanged kernels, changed dispatch), timings of the two schedules
set -u
OUT=$PWD/gpurun_out/r05_wave; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_tree_wave_gpu.py tests/test_tree_engine_gpu.py -q 2>&1 | grep -v "$F" | tail -30 | tee "$OUT/pytest_wave.txt"
timeout 600 python scripts/time_tree_wave.py 2>&1 | grep -v "$F" | tee "$OUT/time_wave.txt"
