#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_m; mkdir -p "$OUT"
export RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_seedcount.so
python scripts/time_c3.py 10000 500 2>&1 | grep "frozen from" | sort | uniq -c | sort -rn | head -30 | tee "$OUT/frozen_L20.txt"
python scripts/time_c3.py 10000 0 2>&1 | grep "frozen from" | sort | uniq -c | sort -rn | head -20 | tee "$OUT/frozen_L10.txt"
