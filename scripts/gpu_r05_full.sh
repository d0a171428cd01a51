#!/bin/bash
# the whole -m gpu suite, smoke, the bench line
set -u
OUT=$PWD/gpurun_out/r05_full; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -25 | tee "$OUT/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -6 | tee "$OUT/smoke.txt"
timeout 900 python bench.py 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"; tail -3 "$OUT/bench.err"
python - <<'PY'
import json
p = json.load(open("gpurun_out/r05_full/bench.json"))
x = p["extra"]
print("ms_per_step", p["ms_per_step"], "roofline", p["roofline"]["frac"], "c3", x["c3"]["ms_per_step"], x["c3"].get("parity_spot"), "c2_missing", x["c2_missing"]["ms_per_step"])
print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in x.items()})
PY
