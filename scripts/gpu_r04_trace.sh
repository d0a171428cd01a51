#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_trace; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | tee "$OUT/pytest.txt"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline 2>"$OUT/bench$i.err" | tail -1 > "$OUT/bench$i.json"
python - $i <<'PY'
import json, sys
p = json.load(open(f"gpurun_out/r04_trace/bench{sys.argv[1]}.json"))
print("ms_per_step", p["ms_per_step"], "c3", p["extra"]["c3"]["ms_per_step"], p["extra"]["c3"]["create_set_data_first_run_ms"], p["extra"]["c3"]["create_stages_ms"])
print("d64", p["extra"]["mid_sizes"]["d64_chains64_T1000"]["create_set_data_first_run_ms"], "d8", p["extra"]["mid_sizes"]["d8_chains1024_T1000"]["create_set_data_first_run_ms"], "c1", p["extra"]["c1"]["infer_ms"])
PY
done
