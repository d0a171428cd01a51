#!/bin/bash
# PMC pass for the HGF kernel (BASELINE config 4): instruction mix and issue/wait split; run via gpurun
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c4pmc; mkdir -p "$OUT"
CMD="python $PWD/scripts/prof_driver.py --config c4 --steps 2 --warmup 1"
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d "$OUT/a" -o a -- $CMD > /dev/null 2> "$OUT/a.err"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/b" -o b -- $CMD > /dev/null 2> "$OUT/b.err"
cd - > /dev/null
python3 scripts/summarize_prof.py "$OUT" 2>&1 | grep -v "^== kernel stats"
tail -3 "$OUT/b.err"
find "$OUT" -name "*.csv" -size +4M -delete
