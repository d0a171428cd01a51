#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03i; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_device_tables_gpu.py -m gpu -x -q -s 2>&1 | tail -25 | tee "$OUT/pytest_tab.txt"
RXHIP_TRACE=1 python scripts/time_create_c3.py 2>&1 | tail -40 | tee "$OUT/create_c3.txt"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee "$OUT/pytest.txt"
python scripts/prof_driver.py --config c3 --steps 30 --warmup 3 2>&1 | tail -1 | tee "$OUT/driver_c3.txt"
