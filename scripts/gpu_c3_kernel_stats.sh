#!/bin/bash
# kernel stats of the C3 sweep only (quick look while tuning one kernel)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/c3_stats; mkdir -p "$OUT"; ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3" -o c3 -- python $ROOT/scripts/prof_driver.py --config c3 --steps 10 --warmup 2 > "$OUT/driver.txt" 2> "$OUT/err.txt"
cd "$ROOT"
f=$(find "$OUT/c3" -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4,6,7 "$f" | head -24 | tee "$OUT/stats.txt"
find "$OUT" -name "*.csv" -size +4M -delete
