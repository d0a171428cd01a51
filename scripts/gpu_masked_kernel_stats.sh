#!/bin/bash
# kernel stats of the masked time-parallel schedule:  ./scripts/gpu_masked_kernel_stats.sh ["d chains T" ...]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/masked_stats; mkdir -p "$OUT"; ROOT=$PWD
cd /tmp
if [ $# -eq 0 ]; then set -- "64 1 2000" "8 1024 1000"; fi
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -o m -- python $ROOT/scripts/prof_mseg.py $cfg > "$OUT/$tag.txt" 2> "$OUT/$tag.err"
  cat "$OUT/$tag.txt"
  f=$(find "$OUT/$tag" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s} %")
PY
done
find "$OUT" -name "*.csv" -size +4M -delete
