#!/bin/bash
# the -m gpu suite and one bench line (what the driver runs at round end)
set -u
OUT=$PWD/gpurun_out/r04_full; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | tee "$OUT/pytest.txt"
timeout 600 python bench.py 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"; tail -2 "$OUT/bench.err"
python - <<'PY'
import json
p = json.load(open("gpurun_out/r04_full/bench.json"))
e = p["extra"]
print("headline ms", p["ms_per_step"], "frac", p["roofline"]["frac"], "parity", p["parity_spot"]["ok"], "cpu all-cores", p["cpu_baseline"]["all_cores"]["value"] / p["cpu_baseline"]["value"])
print("per-chain", p["roofline_per_chain_models"]["ms_per_step"], "c2_missing", e["c2_missing"]["ms_per_step"])
print("c1", e["c1"]["infer_ms"], "c3", e["c3"]["ms_per_step"], e["c3"]["kernels_ms_avg"], e["c3"]["create_set_data_first_run_ms"], e["c3"]["roofline"]["mfma_frac"])
print("c4", e["c4"]["ms_per_step"], "c5", e["c5"]["ms_per_iteration"])
for k, v in e["mid_sizes"].items(): print(k, v["ms_per_step"], v["create_set_data_first_run_ms"], v["covariances_on_request"]["ms_per_step"], v["parity_spot"]["ok"])
print("masked", e["masked_mfma"])
for k in ("c3", "c4", "c5"): print(k, e[k].get("parity_spot"))
PY
