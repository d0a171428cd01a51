#!/bin/bash
# A/B of kernel variants on one box: every rxinfer.jl_amd/csrc/variants/librxhip_*.so runs the C3 sweep (prof_driver) through RXHIP_LIB
set -u
OUT=$PWD/gpurun_out/${1:-ab}
mkdir -p "$OUT"
for rep in 1 2; do
for so in rxinfer.jl_amd/csrc/variants/librxhip_*.so; do
  name=$(basename "$so" .so)
  echo "== $name (run $rep)" | tee -a "$OUT/ab_c3.txt"
  RXHIP_LIB=$PWD/$so python scripts/prof_driver.py --config c3 --steps 30 --warmup 3 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tee -a "$OUT/ab_c3.txt"
done
done
