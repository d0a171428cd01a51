#!/bin/bash
# PMC pass for the dense d = 64 path (BASELINE config 3): MFMA / fp64 instruction mix and LDS behaviour per kernel; run via gpurun
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c3pmc; mkdir -p "$OUT"
CMD="python $PWD/scripts/prof_driver.py --config c3 --steps 3 --warmup 1"
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/a" -o a -- $CMD > /dev/null 2> "$OUT/a.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/b" -o b -- $CMD > /dev/null 2> "$OUT/b.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/c" -o c -- $CMD > /dev/null 2> "$OUT/c.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/d" -o d -- $CMD > /dev/null 2> "$OUT/d.err"
cd - > /dev/null
python3 scripts/summarize_prof.py "$OUT" 2>&1 | grep -v "^== kernel stats"
find "$OUT" -name "*.csv" -size +4M -delete
