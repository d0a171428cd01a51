#!/bin/bash
# Run on the GPU box (via gpurun).  Round-5 evidence under gpurun_out/prof_r05 (copied to profiles/r05/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (headline workload only)                          -> kernel_stats_bench.csv
#   2. PMC FETCH_SIZE / WRITE_SIZE passes (separate, MI355X_MICROARCH.md) on the C2 driver                   -> traffic.json
#   3. C3: kernel stats + two PMC passes (MFMA instructions executed; waits, matrix-pipe busy)               -> kernel_stats_c3.csv, pmc_c3.txt, mfma_insts.json
#   4. the node-array executor (bench's node_array workload): kernel stats + FETCH_SIZE / WRITE_SIZE passes  -> kernel_stats_tree.csv, tree_traffic.json
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.err"
DRV="python $ROOT/scripts/prof_driver.py --steps 5 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $DRV > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $DRV > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3" -o c3 -- python $ROOT/scripts/prof_driver.py --config c3 --steps 10 --warmup 2 > "$OUT/driver_c3.txt" 2> "$OUT/c3.err"
C3="python $ROOT/scripts/prof_driver.py --config c3 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/c3pmc_a" -o a -- $C3 > /dev/null 2> "$OUT/c3pmc_a.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$OUT/c3pmc_c" -o c -- $C3 > /dev/null 2> "$OUT/c3pmc_c.err"
TREE="python $ROOT/scripts/prof_tree.py 128 65536 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tree" -o tree -- $TREE > "$OUT/driver_tree.txt" 2> "$OUT/tree.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/tree_fetch" -o fetch -- $TREE > /dev/null 2> "$OUT/tree_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/tree_write" -o write -- $TREE > /dev/null 2> "$OUT/tree_write.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$OUT/tree_pmc" -o t -- $TREE > /dev/null 2> "$OUT/tree_pmc.err"
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, hashlib, json, os, sys
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
sha = lambda f: hashlib.sha256(open(os.path.join("rxinfer.jl_amd", "csrc", f), "rb").read()).hexdigest()
c2 = ("k_seg_aggregate", "k_boundary_scan", "k_forward", "k_backward")
fetch, write = avg("FETCH_SIZE", "pmc_fetch", c2), avg("WRITE_SIZE", "pmc_write", c2)
t = {"source": f"profiles/{tag}/ (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, scripts/prof_driver.py C2 workload, shared-model batch)",
     "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced streaming reads (MI355X_MICROARCH.md HBM section); KiB -> bytes x1024"}
for k in ("k_seg_aggregate", "k_forward", "k_backward"):
    if k in fetch and k in write:
        t[f"{k}_fetch_bytes_per_launch"] = fetch[k] * 1024 * 2
        t[f"{k}_write_bytes_per_launch"] = write[k] * 1024
        t[f"{k}_hbm_bytes_per_launch"] = fetch[k] * 1024 * 2 + write[k] * 1024
t["lgssm_kernels_sha256"] = sha("lgssm_kernels.hpp")
json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
# C3: MFMA instructions the sweep kernels EXECUTE per launch (v_mfma_f64_16x16x4_f64 = 2048 flop each)
kern = ("kd_forward_info", "kd_backward_info", "kd_fe_resid_mfma", "kd_agg_gemm", "kd_scan_fix", "kd_scan_local")
m = {"source": f"profiles/{tag}/ (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 …, scripts/prof_driver.py --config c3: d = dy = 64, T = 10^4, one chain; average per dispatch over the timed sweeps)",
     "flop_per_instruction": 2048, "mfma_f64_per_launch": avg("SQ_INSTS_VALU_MFMA_F64", "c3pmc_a", kern, last=3),
     "mfma_busy_cycles_per_launch": avg("SQ_VALU_MFMA_BUSY_CYCLES", "c3pmc_c", kern, last=3), "busy_cycles_per_launch": avg("SQ_BUSY_CYCLES", "c3pmc_a", kern, last=3),
     "dense_kernels_sha256": sha("dense_kernels.hpp")}
json.dump(m, open(os.path.join(out, "mfma_insts.json"), "w"), indent=1)
# the node-array executor: HBM bytes per launch of its kernel instances (the format bench.py reads: profiles/tree_traffic.json)
tk = ("k_tree_levels<4, 0>", "k_tree_levels<4, 1>", "k_tree_walk<4, 0>", "k_tree_walk<4, 1>", "k_tree_fe_total")   # (whichever schedule the engine picked for the batch)
tf, tw = avg("FETCH_SIZE", "tree_fetch", tk, last=3), avg("WRITE_SIZE", "tree_write", tk, last=3)
alg = None
try:
    drv = [l for l in open(os.path.join(out, "driver_tree.txt")) if l.startswith("{")][-1]
    info = eval(drv)   # (the driver prints a Python dict)
    alg = info["info"]["bytes_per_sweep"] * info["replicas"]
except Exception:
    pass
tt = {"source": f"profiles/{tag}/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, scripts/prof_tree.py 128 65536 3: the bench's node_array workload — two observation branches per state, d = 4, T = 128, 65 536 replicas, the schedule the engine picks for that batch; average of the last 3 launches of each kernel instance)",
      "correction": "KiB units x1024; FETCH_SIZE as reported and x2 (the guide's gfx950 correction is for 16 B/lane streams; the executor loads 8 B/lane unit-stride): both given",
      "algorithmic_bytes_per_sweep": alg, "tree_kernels_sha256": sha("tree_kernels.hpp"), "kernels": {}}
for k in tk:
    if k in tf and k in tw:
        tt["kernels"][k] = {"fetch_bytes_x1": tf[k] * 1024, "fetch_bytes_x2": tf[k] * 2048, "write_bytes": tw[k] * 1024,
                            "hbm_bytes_per_launch_x1": tf[k] * 1024 + tw[k] * 1024, "hbm_bytes_per_launch_x2": tf[k] * 2048 + tw[k] * 1024}
json.dump(tt, open(os.path.join(out, "tree_traffic.json"), "w"), indent=1)
print(json.dumps(t, indent=1)); print(json.dumps(m, indent=1)); print(json.dumps(tt, indent=1))
PY
cat "$OUT/summary.txt" | cut -c1-400
cat "$OUT/driver_tree.txt" "$OUT/driver_c3.txt" | grep -v 'RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
for d in bench c3 tree; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$d.csv"; done
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*_agent_info.csv" -delete
