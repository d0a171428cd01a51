#!/bin/bash
# kernel stats of mid-size shared-model batches
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/mid_stats; mkdir -p "$OUT"; ROOT=$PWD
cd /tmp
for cfg in "8 8 1024 1000" "64 64 64 1000"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -o m -- python $ROOT/scripts/prof_mid.py $cfg > "$OUT/$tag.txt" 2> "$OUT/$tag.err"
  cat "$OUT/$tag.txt"
  f=$(find "$OUT/$tag" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s} %")
PY
done
find "$OUT" -name "*.csv" -size +4M -delete
