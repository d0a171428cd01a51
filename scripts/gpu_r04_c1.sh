#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_c1; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 600 python -m pytest tests/test_small_sweep_gpu.py tests/test_lgssm_gpu.py tests/test_lgssm_filter_gpu.py tests/test_predictions_gpu.py tests/test_known_inputs.py tests/test_time_varying.py -x -q 2>&1 | grep -v "$F" | tail -8 | tee "$OUT/pytest.txt"
python scripts/time_c1_breakdown.py 2>&1 | grep -v "$F" | tee "$OUT/c1_breakdown.txt"
python scripts/notebook_sizes.py 2>&1 | grep -v "$F" | tee "$OUT/notebook_sizes.txt"
