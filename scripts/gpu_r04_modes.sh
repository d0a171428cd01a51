#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_modes; mkdir -p "$OUT"
for k in 1 2 3 4 5; do MODES_PLAIN=1 timeout 300 python scripts/time_placement_modes.py 2>&1 | grep "default stream" | tee -a "$OUT/modes.txt"; done
