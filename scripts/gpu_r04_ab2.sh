#!/bin/bash
# headline line of bench.py under library variants (RXHIP_LIB); usage: gpu_r04_ab2.sh name1 name2 ...
set -u
OUT=$PWD/gpurun_out/r04_ab; mkdir -p "$OUT"
for v in "$@"; do
  if [ "$v" = default ]; then unset RXHIP_LIB; else export RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_$v.so; fi
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernels_ms_avg'])" | tee -a "$OUT/ab2.txt"
done
