#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_c3; mkdir -p "$OUT"; rm -f "$OUT/c3_ab.txt"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
for rep in 1 2; do
for so in rxinfer.jl_amd/csrc/variants/librxhip_*.so; do
RXHIP_LIB=$PWD/$so python scripts/time_c3_clean.py 2>&1 | grep -v "$F" | tail -1 | tee -a "$OUT/c3_ab.txt"
done
done
python scripts/debug_seed.py 2>&1 | grep -v "$F" | cut -c1-250
