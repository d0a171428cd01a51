#!/bin/bash
set -u
OUT=$PWD/gpurun_out/tests_c3; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | tail -5 | tee "$OUT/pytest.txt"
for i in 1 2; do python scripts/prof_driver.py --config c3 --steps 30 --warmup 3 2>&1 | tail -1 | tee -a "$OUT/driver_c3.txt"; done
python scripts/time_dense_split.py 2>&1 | tail -12 | tee "$OUT/dense_split.txt"
