#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats, then PMC passes (separately,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes), all on the same bench.py command.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/scripts/prof_driver.py --steps 5 --warmup 2 ${PROF_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $CMD > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -o sq -- $CMD > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_mem" -o mem -- $CMD > /dev/null 2> "$OUT/pmc_mem.err"
cd - > /dev/null
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small files for the merge back
find "$OUT" -name "*.csv" -size +8M -delete
