#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_n; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|create +'
for seg in 0 1112 1250 1429 1667 2000 2500; do
  python scripts/time_c3.py 10000 $seg 2>&1 | grep -v "$F" | tee -a "$OUT/c3_short_segments.txt"
done
