#!/bin/bash
# second PMC pass for C3: where the wave cycles of the sweep kernels go (waits by kind, issue by kind)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c3pmc2; mkdir -p "$OUT"
CMD="python $PWD/scripts/prof_driver.py --config c3 --steps 3 --warmup 1"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/a" -o a -- $CMD > /dev/null 2> "$OUT/a.err"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT/b" -o b -- $CMD > /dev/null 2> "$OUT/b.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_MISC --output-format csv -d "$OUT/c" -o c -- $CMD > /dev/null 2> "$OUT/c.err"
cd - > /dev/null
python3 scripts/summarize_prof.py "$OUT" 2>&1 | grep -v "^== kernel stats"
tail -3 "$OUT"/*.err
find "$OUT" -name "*.csv" -size +4M -delete
