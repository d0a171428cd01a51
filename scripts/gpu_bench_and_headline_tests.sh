#!/bin/bash
set -u
OUT=$PWD/gpurun_out/bench_headline; mkdir -p "$OUT"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -3 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(x, ind=0):
    for k,v in x.items():
        if isinstance(v, dict): print(" "*ind+k+":"); show(v, ind+2)
        else: print(" "*ind+f"{k}: {v if not isinstance(v,str) else v[:110]}")
show(d)
PY
timeout 1200 python -m pytest tests/test_headline_parity_gpu.py tests/test_bench_contract_gpu.py tests/test_lgssm_gpu.py tests/test_random_shapes_gpu.py tests/test_known_inputs.py -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/pytest.txt"
