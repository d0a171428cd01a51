#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_seg; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
python scripts/time_split_segments.py 2>&1 | grep -v "$F" | tee "$OUT/split_segments.txt"
RXHIP_TRACE=1 python scripts/time_create_c3.py 2>&1 | grep -v "$F" | grep 'model:' | tee "$OUT/create_c3.txt"
