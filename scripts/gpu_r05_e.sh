#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_e; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
for v in tab14 tab5e15 tab1e15; do
  if [ $v = default ]; then unset RXHIP_LIB; else export RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_$v.so; fi
  python scripts/time_c3.py 2>&1 | grep -v "$F" | tee -a "$OUT/time_c3.txt"
done
