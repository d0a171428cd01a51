#!/bin/bash
# round 3, call 1: the blocked inverse in isolation, the whole GPU suite on it, C3 kernel breakdown
set -u
OUT=$PWD/gpurun_out/r03a
mkdir -p "$OUT"
./scripts/inv_micro > "$OUT/inv_micro.txt" 2>&1
tail -60 "$OUT/inv_micro.txt"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest.txt"
python scripts/prof_driver.py --config c3 --steps 20 --warmup 3 2>&1 | tee "$OUT/driver_c3.txt"
RXHIP_DENSE_SPLIT=0 python scripts/time_mid_dims.py quick 2>&1 | tee "$OUT/mid_dims.txt"
