#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03f; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee "$OUT/pytest.txt"
python scripts/prof_driver.py --config c3 --steps 30 --warmup 3 2>&1 | tail -1 | tee "$OUT/driver_c3.txt"
RXHIP_FE_RESID_VALU=1 python scripts/prof_driver.py --config c3 --steps 30 --warmup 3 2>&1 | tail -1 | tee -a "$OUT/driver_c3.txt"
python scripts/time_dense_split.py 2>&1 | tail -12 | tee "$OUT/dense_split.txt"
