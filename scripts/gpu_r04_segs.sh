#!/bin/bash
# BASELINE config 3 with other segment counts (workgroups per CU of the forward/backward kernels: 256 CUs)
set -u
OUT=$PWD/gpurun_out/r04_ab; mkdir -p "$OUT"
for s in "$@"; do
  C3_SEGMENTS=$s RXHIP_TEST_HOOKS=1 timeout 300 python scripts/time_c3_clean.py 2>&1 | grep -v 'amdgpu.ids' | tee -a "$OUT/segs.txt"
done
