#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_trace; mkdir -p "$OUT"
RXHIP_TRACE=1 timeout 600 python bench.py --no-cpu-baseline 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"
grep -n 'rxhip' "$OUT/bench.err" | tail -150
