#!/bin/bash
# Run on the GPU box (via gpurun).  Round-4 evidence, everything under gpurun_out/prof_r03 (copied to profiles/r04/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (headline workload only)            -> kernel_stats_bench.csv
#   2. PMC FETCH_SIZE / WRITE_SIZE passes (separate, MI355X_MICROARCH.md) on the C2 driver     -> traffic.json
#   3. C3: kernel stats + two PMC passes (MFMA / VALU mix; waits, LDS, issue)                  -> kernel_stats_c3.csv, pmc_c3.txt
#   4. C4, C5: kernel stats + PMC (VALU instructions, wave cycles)                             -> kernel_stats_c4/5.csv, valu_insts.json
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.err"
DRV="python $ROOT/scripts/prof_driver.py --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $DRV > "$OUT/trace_driver.txt" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $DRV > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $DRV > /dev/null 2> "$OUT/pmc_write.err"
for cfg in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$cfg" -o $cfg -- python $ROOT/scripts/prof_driver.py --config $cfg --steps 10 --warmup 2 > "$OUT/driver_$cfg.txt" 2> "$OUT/$cfg.err"
done
C3="python $ROOT/scripts/prof_driver.py --config c3 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/c3pmc_a" -o a -- $C3 > /dev/null 2> "$OUT/c3pmc_a.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/c3pmc_b" -o b -- $C3 > /dev/null 2> "$OUT/c3pmc_b.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$OUT/c3pmc_c" -o c -- $C3 > /dev/null 2> "$OUT/c3pmc_c.err"
for cfg in c4 c5; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$OUT/${cfg}pmc" -o $cfg -- python $ROOT/scripts/prof_driver.py --config $cfg --steps 2 --warmup 1 > /dev/null 2> "$OUT/${cfg}pmc.err"
done
# round 4: masked d <= 8 batches (in-wave kernels), the composed graph (noise VMP), small problems, first touch
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/w8" -o w8 -- python $ROOT/scripts/prof_masked8.py > "$OUT/driver_masked8.txt" 2> "$OUT/w8.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d "$OUT/w8pmc" -o w8 -- python $ROOT/scripts/prof_masked8.py > /dev/null 2> "$OUT/w8pmc.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/noise" -o noise -- python $ROOT/scripts/prof_noise.py > "$OUT/driver_noise.txt" 2> "$OUT/noise.err"
cd "$ROOT"
python scripts/time_masked_small.py 2>&1 | grep -v "$F" > "$OUT/masked_small.txt"
python scripts/time_c1_breakdown.py 2>&1 | grep -v "$F" > "$OUT/c1_breakdown.txt"
python scripts/notebook_sizes.py 2>&1 | grep -v "$F" > "$OUT/notebook_sizes.txt"
python scripts/time_split_segments.py 2>&1 | grep -v "$F" > "$OUT/split_segments.txt"
RXHIP_TRACE=1 python scripts/time_create_c3.py 2>&1 | grep -v "$F" > "$OUT/create_c3_trace.txt"
python scripts/time_c3_clean.py 2>&1 | grep -v "$F" | tail -1 > "$OUT/c3_clean.txt"
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
def avg(counter, sub, names, last=5):
    vals = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"].split("(")[0]
            for k in names:
                if k in n:
                    vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v[-last:]) / len(v[-last:]) for k, v in vals.items()}
c2 = ("k_seg_aggregate", "k_boundary_scan", "k_forward", "k_backward")
fetch, write = avg("FETCH_SIZE", "pmc_fetch", c2), avg("WRITE_SIZE", "pmc_write", c2)
t = {"source": f"profiles/{tag}/ (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, scripts/prof_driver.py C2 workload, shared-model batch)",
     "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced streaming reads (MI355X_MICROARCH.md HBM section); KiB -> bytes x1024"}
for k in ("k_seg_aggregate", "k_forward", "k_backward"):
    if k in fetch and k in write:
        t[f"{k}_fetch_bytes_per_launch"] = fetch[k] * 1024 * 2
        t[f"{k}_write_bytes_per_launch"] = write[k] * 1024
        t[f"{k}_hbm_bytes_per_launch"] = fetch[k] * 1024 * 2 + write[k] * 1024
import hashlib
# the kernels these counters belong to: bench.py flags the figure as stale when the source no longer hashes to this
t["lgssm_kernels_sha256"] = hashlib.sha256(open(os.path.join("rxinfer.jl_amd", "csrc", "lgssm_kernels.hpp"), "rb").read()).hexdigest()
json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
v = {"source": f"profiles/{tag}/ (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES …, scripts/prof_driver.py --config c4 / c5; per launch = per filtering pass of 4096 series x 2000 observations x 10 iterations (c4), per VMP iteration over 1e7 points (c5))"}
for cfg, kern in (("c4", "k_hgf_filter"), ("c5", "k_gmm_pass")):
    v[cfg] = {"kernel": kern}
    for ctr in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVES"):
        a = avg(ctr, cfg + "pmc", (kern,), last=3)
        if kern in a:
            v[cfg][ctr] = a[kern]
json.dump(v, open(os.path.join(out, "valu_insts.json"), "w"), indent=1)
print(json.dumps(t, indent=1)); print(json.dumps(v, indent=1))
PY
cat "$OUT/summary.txt"
for d in bench trace c3 c4 c5 w8 noise; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$d.csv"; done
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*_agent_info.csv" -delete
