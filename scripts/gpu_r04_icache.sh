#!/bin/bash
# instruction-cache counters of the C3 sweep kernels (are the 30 KB loops of kd_forward_info fetch-bound?)
set -u
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r04_icache; mkdir -p "$OUT"
C3="python $ROOT/scripts/prof_driver.py --config c3 --steps 4 --warmup 2"
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u > "$OUT/counters.txt"
cat "$OUT/counters.txt"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d "$OUT/ic" -o ic -- $C3 > /dev/null 2> "$OUT/ic.err"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "kd_" in k:
        print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
