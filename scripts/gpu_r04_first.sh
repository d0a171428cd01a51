#!/bin/bash
# round 4, first box call: the split library under the -m gpu suite, the bench line, and what a first touch is made of
set -u
OUT=$PWD/gpurun_out/r04_first; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | tee "$OUT/pytest.txt"
./scripts/scratch_first_touch 2>&1 | tee "$OUT/scratch_first_touch.txt"
for i in 1 2; do RXHIP_TRACE=1 python scripts/time_create_c3.py 2>&1 | grep -v "$F" | tee -a "$OUT/create_c3_trace.txt"; echo ---- >> "$OUT/create_c3_trace.txt"; done
timeout 600 python bench.py 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"; tail -3 "$OUT/bench.err"
python - <<'PY'
import json
p = json.load(open("gpurun_out/r04_first/bench.json"))
print("ms_per_step", p["ms_per_step"], "c3", p["extra"]["c3"]["ms_per_step"], p["extra"]["c3"]["create_set_data_first_run_ms"], p["extra"]["c3"]["create_stages_ms"])
print("d64", p["extra"]["mid_sizes"]["d64_chains64_T1000"]["create_set_data_first_run_ms"], "c1", p["extra"]["c1"]["infer_ms"])
PY
