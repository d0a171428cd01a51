#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03g; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee "$OUT/pytest.txt"
python scripts/prof_driver.py --config c3 --steps 30 --warmup 3 2>&1 | tail -1 | tee "$OUT/driver_c3.txt"
RXHIP_DENSE_SPLIT=0 python scripts/time_mid_dims.py quick 2>&1 | tee "$OUT/mid_dims.txt"
