#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_i; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|create +'
for seg in 0 1000 768 640 512 500 400 333 256 200; do
  python scripts/time_c3.py 10000 $seg 2>&1 | grep -v "$F" | tee -a "$OUT/c3_segments.txt"
done
