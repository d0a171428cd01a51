#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03s; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_dense_missing_parallel_gpu.py -m gpu -x -q 2>&1 | tail -25 | tee "$OUT/pytest_mseg.txt"
timeout 900 python -m pytest tests/test_dense_sequential.py tests/test_known_inputs.py tests/test_missing_observations.py tests/test_predictions_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee "$OUT/pytest2.txt"
timeout 600 python scripts/time_dense_sequential.py 2>&1 | tail -8 | tee "$OUT/dense_missing.txt"
