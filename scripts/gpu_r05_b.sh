#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_b; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
python scripts/time_c3.py 2>&1 | grep -v "$F" | tee "$OUT/time_c3.txt"
RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_r4base.so python scripts/time_c3.py 2>&1 | grep -v "$F" | tee -a "$OUT/time_c3.txt"
timeout 900 python scripts/diag_fixed_point.py 2>&1 | grep -v "$F" | tee "$OUT/diag_fixed_point.txt"
