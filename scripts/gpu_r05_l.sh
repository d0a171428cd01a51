#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_l; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|create +'
for v in default lane3e-14 lane6e-14 lane1.1e-13; do
  if [ $v = default ]; then unset RXHIP_LIB; else export RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_$v.so; fi
  for seg in 0 500 250; do
    python scripts/time_c3.py 10000 $seg 2>&1 | grep -v "$F" | sed "s|^lib [^ ]*|$v|" | tee -a "$OUT/c3_lane_tol.txt"
  done
done
