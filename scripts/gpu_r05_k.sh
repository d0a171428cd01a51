#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_k; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|create +'
for seg in 0 500 250; do
  python scripts/time_c3.py 10000 $seg 2>&1 | grep -v "$F" | tee -a "$OUT/c3_nofrozen.txt"
  RXHIP_TEST_HOOKS=1 RXHIP_NO_FROZEN=1 python scripts/time_c3.py 10000 $seg 2>&1 | grep -v "$F" | sed 's/^lib default/NO_FROZEN/' | tee -a "$OUT/c3_nofrozen.txt"
done
