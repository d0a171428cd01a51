#!/bin/bash
# round 5, first box call: the fixed-point tests (adversarial inputs) on the new criteria, the sweeps that freeze, and three bench lines —
# this tree, the same tree with non-temporal posterior stores in k_backward_sh (variants/librxhip_nt.so), round 4's library (variants/librxhip_r4base.so)
set -u
OUT=$PWD/gpurun_out/r05_a; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 900 python -m pytest tests/test_fixed_point_adversarial_gpu.py tests/test_seeded_tile_inverse_gpu.py tests/test_converged_elements_gpu.py tests/test_badly_scaled_models_gpu.py tests/test_headline_parity_gpu.py -m gpu -q 2>&1 | grep -v "$F" | tail -40 | tee "$OUT/pytest_fixed_point.txt"
echo "---- adversarial tests against round 4's library (expected to FAIL where the old criterion is blind)" | tee -a "$OUT/pytest_fixed_point.txt"
RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_r4base.so timeout 600 python -m pytest tests/test_fixed_point_adversarial_gpu.py -m gpu -q 2>&1 | grep -v "$F" | tail -25 | tee "$OUT/pytest_fixed_point_r4lib.txt"
for v in cur nt r4base; do
  if [ $v = cur ]; then unset RXHIP_LIB; else export RXHIP_LIB=$PWD/rxinfer.jl_amd/csrc/variants/librxhip_$v.so; fi
  timeout 600 python bench.py 2>"$OUT/bench_$v.err" | tail -1 > "$OUT/bench_$v.json"; tail -2 "$OUT/bench_$v.err"
done
unset RXHIP_LIB
python - <<'PY'
import json
for v in ("cur", "nt", "r4base"):
    try:
        p = json.load(open(f"gpurun_out/r05_a/bench_{v}.json"))
        x = p["extra"]
        print(v, "ms_per_step", p["ms_per_step"], "roofline", p["roofline"]["frac"], "kernels", {k: round(t, 3) for k, t in p.get("kernel_ms", {}).items()} if "kernel_ms" in p else "",
              "c3", x["c3"]["ms_per_step"], "c2_missing", x.get("c2_missing", {}).get("ms_per_step"), "per_chain", x.get("c2_per_chain_models", {}).get("ms_per_step"))
    except Exception as e:
        print(v, "failed", e)
PY
