#!/bin/bash
# the -m gpu suite and one bench line (what the driver runs at round end)
set -u
OUT=$PWD/gpurun_out/r04_full; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | tee "$OUT/pytest.txt"
timeout 600 python bench.py 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"; tail -2 "$OUT/bench.err"
python scripts/show_bench.py "$OUT/bench.json"
