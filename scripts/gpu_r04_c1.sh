#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_c1; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
python scripts/time_small_segments.py 2>&1 | grep -v "$F" | tee "$OUT/small_segments.txt"
