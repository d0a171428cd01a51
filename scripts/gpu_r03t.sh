#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03t; mkdir -p "$OUT"
for i in 1 2; do python scripts/prof_driver.py --config c3 --steps 30 --warmup 3 2>&1 | tail -1 | tee -a "$OUT/driver_c3.txt"; done
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python $OLDPWD/scripts/prof_driver.py --config c3 --steps 10 --warmup 2 > /dev/null 2>&1; cd $OLDPWD
python scripts/summarize_prof.py "$OUT" 2>&1 | grep "^kd_\|^k_" | tee "$OUT/kstats.txt"
timeout 900 python -m pytest tests/test_random_shapes_gpu.py tests/test_lgssm_gpu.py tests/test_device_tables_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/pytest.txt"
