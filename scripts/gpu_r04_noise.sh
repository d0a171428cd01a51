#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_noise; mkdir -p "$OUT"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 600 python -m pytest tests/test_noise_vmp_gpu.py -x -q 2>&1 | grep -v "$F" | tail -5 | tee "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o noise -- python scripts/prof_noise.py 2>/dev/null | tail -1 | tee "$OUT/driver_noise.txt"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04_noise/prof/**/*kernel_stats.csv", recursive=True)[0]
import shutil; shutil.copy(f, "gpurun_out/r04_noise/kernel_stats_noise.csv")
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'].split('(')[0][:50]:50s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
