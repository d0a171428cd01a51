#!/bin/bash
# A/B of library variants on BASELINE config 3 (scripts/time_c3_clean.py); usage: gpu_r04_ab.sh name1 name2 ... ("default" = the shipped library)
set -u
OUT=$PWD/gpurun_out/r04_ab; mkdir -p "$OUT"
for v in "$@"; do
  if [ "$v" = default ]; then unset RXHIP_LIB; else export RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_$v.so; fi
  RXHIP_TEST_HOOKS=1 timeout 300 python scripts/time_c3_clean.py 2>&1 | grep -v 'amdgpu.ids' | tee -a "$OUT/ab.txt"
done
