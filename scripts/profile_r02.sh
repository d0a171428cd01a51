#!/bin/bash
# Run on the GPU box (via gpurun).  Round-2 evidence:
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (headline workload only) -> kernel_stats_bench.csv
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE — separately, as MI355X_MICROARCH.md prescribes) on the torch-free driver of the
#      same workload (scripts/prof_driver.py) -> per-kernel HBM bytes -> traffic.json
#   3. kernel stats of C3 / C4 / C5 (prof_driver --config)
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.err"
DRV="python $ROOT/scripts/prof_driver.py --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $DRV > "$OUT/trace_driver.txt" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $DRV > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $DRV > /dev/null 2> "$OUT/pmc_write.err"
for cfg in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$cfg" -o $cfg -- python $ROOT/scripts/prof_driver.py --config $cfg --steps 10 --warmup 2 > "$OUT/driver_$cfg.txt" 2> "$OUT/$cfg.err"
done
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
def avg(counter, sub):
    vals = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"].split("(")[0]
            for k in ("k_seg_aggregate", "k_boundary_scan", "k_forward", "k_backward"):
                if k in n:
                    vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v[-5:]) / len(v[-5:]) for k, v in vals.items()}
fetch, write = avg("FETCH_SIZE", "pmc_fetch"), avg("WRITE_SIZE", "pmc_write")
t = {"source": f"profiles/{os.path.basename(out).replace('prof_', '')}/ (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, scripts/prof_driver.py C2 workload, shared-model batch)",
     "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced streaming reads (MI355X_MICROARCH.md HBM section); KiB -> bytes x1024"}
for k in ("k_seg_aggregate", "k_forward", "k_backward"):
    if k in fetch and k in write:
        t[f"{k}_fetch_bytes_per_launch"] = fetch[k] * 1024 * 2
        t[f"{k}_write_bytes_per_launch"] = write[k] * 1024
        t[f"{k}_hbm_bytes_per_launch"] = fetch[k] * 1024 * 2 + write[k] * 1024
json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(t, indent=1))
PY
cat "$OUT/summary.txt"
for d in bench trace c3 c4 c5; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$d.csv"; done
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*_agent_info.csv" -delete
