#!/bin/bash
# kernel-level view of the table builders inside the bench process: every kt_* / kd_prepare_bnd dispatch with start time and duration
set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=$PWD/gpurun_out/r04_kt; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o bench -- python bench.py --no-cpu-baseline --no-parity > "$OUT/bench.json" 2>"$OUT/bench.err"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04_kt/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
out = open("gpurun_out/r04_kt/kt_dispatches.txt", "w")
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if "kt_" in n or "kd_prepare_bnd" in n:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        pe = int(rows[i - 1]["End_Timestamp"]) if i else s
        line = f"{(s - t0) / 1e6:12.3f} ms  dur {(e - s) / 1e3:10.1f} us  gap_before {(s - pe) / 1e3:10.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}  {n.split('(')[0][:60]}"
        print(line); out.write(line + "\n")
PY
