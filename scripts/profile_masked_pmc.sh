#!/bin/bash
# PMC pass for the masked time-parallel schedule (dense_mseg_kernels.hpp), d = 64, one chain, T = 2000, 10 % missing: MFMA instructions and
# busy cycles per kernel, HBM bytes per dispatch (FETCH_SIZE / WRITE_SIZE in separate passes); run via gpurun
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_masked_pmc; mkdir -p "$OUT"
CMD="python $PWD/scripts/prof_mseg.py ${1:-64} ${2:-1} ${3:-2000}"
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/a" -o a -- $CMD > /dev/null 2> "$OUT/a.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/c" -o c -- $CMD > /dev/null 2> "$OUT/c.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/d" -o d -- $CMD > /dev/null 2> "$OUT/d.err"
cd - > /dev/null
python3 scripts/summarize_prof.py "$OUT" 2>&1 | grep -v "^== kernel stats"
find "$OUT" -name "*.csv" -size +4M -delete
