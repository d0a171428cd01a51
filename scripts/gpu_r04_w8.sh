#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_w8; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 600 python -m pytest tests/test_wave8_gpu.py tests/test_small_sweep_gpu.py tests/test_node_marginals.py -x -q 2>&1 | grep -v "$F" | tail -5 | tee "$OUT/pytest.txt"
python scripts/time_masked_small.py 2>&1 | grep -v "$F" | tee "$OUT/masked_small.txt"
