#!/bin/bash
# the C3 first touch inside the full bench process, several processes in a row (is the 70 ms outlier a property of the process or of the box?)
set -u
OUT=$PWD/gpurun_out/r04_first3; mkdir -p "$OUT"
for k in 1 2 3 4; do
  RXHIP_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 2>"$OUT/bench$k.err" | tail -1 > "$OUT/bench$k.json"
  python - "$OUT/bench$k.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
c3 = d["extra"]["c3"]
print("c3 first touch", round(c3["create_set_data_first_run_ms"], 2), {k: round(v, 2) for k, v in c3["create_set_data_first_run_split_ms"].items()}, c3["create_stages_ms"],
      "| d64x64", round(d["extra"]["mid_sizes"]["d64_chains64_T1000"]["create_set_data_first_run_ms"], 2))
PY
done
