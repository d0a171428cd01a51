#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_h; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/m64" -o m64 -- python $ROOT/scripts/time_masked64.py 2000 > "$OUT/driver.txt" 2> "$OUT/err.txt"
cd $ROOT
f=$(find "$OUT/m64" -name "*kernel_stats.csv" | head -1); cp "$f" "$OUT/kernel_stats_masked64.csv"
python3 - "$OUT/kernel_stats_masked64.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(r["Name"].split("(")[0][:70].ljust(70), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"].rjust(14), r["Percentage"])
PY
find "$OUT" -name "*.csv" -size +2M -delete
cat "$OUT/driver.txt" | grep -v 'RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
