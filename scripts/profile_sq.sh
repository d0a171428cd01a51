#!/bin/bash
# quick SQ-counter pass: bash scripts/profile_sq.sh <tag> <prof_driver args...>
TAG=$1; shift
OUT=$PWD/gpurun_out/sq_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python $PWD/scripts/prof_driver.py --steps 3 --warmup 1 $*"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/a" -o a -- $CMD > "$OUT/out.txt" 2> "$OUT/a.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/b" -o b -- $CMD > /dev/null 2> "$OUT/b.err"
cd - > /dev/null
python3 scripts/summarize_prof.py "$OUT" | grep -v "^k_fe\|note"
