#!/bin/bash
# Run on the GPU box (via gpurun).  Round-6 evidence under gpurun_out/prof_r06 (copied to profiles/r06/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (headline workload only)                             -> kernel_stats_bench.csv
#   2. PMC FETCH_SIZE / WRITE_SIZE passes (separate, MI355X_MICROARCH.md) on the C2 driver                      -> traffic.json
#   3. C3: kernel stats + PMC (MFMA instructions executed)                                                      -> kernel_stats_c3.csv, mfma_insts.json
#   4. C4 / C5: VALU issue counters                                                                             -> valu_insts.json
#   5. the node-array executor, d = 4 (bench's node_array workload): kernel trace, FETCH_SIZE / WRITE_SIZE,
#      with the calibration of FETCH_SIZE for 8 B/lane unit-stride loads (scripts/fetch_calib.hip)              -> kernel_stats_tree.csv, tree_traffic.json, fetch_calibration.txt
#   6. the executor at d = 64 (work items of four wavefronts, MFMA products and inverse): kernel stats, MFMA instructions  -> kernel_stats_tree64.csv, tree_mfma.json
#   7. the executor's register-tile kernels at d = 16 / 32: kernel stats                                       -> kernel_stats_tree16.csv, kernel_stats_tree32.csv
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o "$OUT/fetch_calib" scripts/fetch_calib.hip 2> "$OUT/fetch_calib.build.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --detail "" > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.err"
DRV="python $ROOT/scripts/prof_driver.py --steps 5 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $DRV > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $DRV > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3" -o c3 -- python $ROOT/scripts/prof_driver.py --config c3 --steps 10 --warmup 2 > "$OUT/driver_c3.txt" 2> "$OUT/c3.err"
C3="python $ROOT/scripts/prof_driver.py --config c3 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/c3pmc_a" -o a -- $C3 > /dev/null 2> "$OUT/c3pmc_a.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d "$OUT/c3pmc_c" -o c -- $C3 > /dev/null 2> "$OUT/c3pmc_c.err"
for cfg in c4 c5; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$OUT/${cfg}pmc" -o $cfg -- python $ROOT/scripts/prof_driver.py --config $cfg --steps 2 --warmup 1 > "$OUT/driver_$cfg.txt" 2> "$OUT/${cfg}pmc.err"
done
TREE="python $ROOT/scripts/prof_tree.py 128 65536 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tree" -o tree -- $TREE > "$OUT/driver_tree.txt" 2> "$OUT/tree.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/tree_fetch" -o fetch -- $TREE > /dev/null 2> "$OUT/tree_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/tree_write" -o write -- $TREE > /dev/null 2> "$OUT/tree_write.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$OUT/tree_pmc" -o t -- $TREE > /dev/null 2> "$OUT/tree_pmc.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -o fetch -- "$OUT/fetch_calib" > "$OUT/fetch_calib.txt" 2> "$OUT/calib_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -o write -- "$OUT/fetch_calib" > /dev/null 2> "$OUT/calib_write.err"
T64="python $ROOT/scripts/prof_tree_wave.py 64 16 256 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tree64" -o tree64 -- $T64 > "$OUT/driver_tree64.txt" 2> "$OUT/tree64.err"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d "$OUT/tree64_pmc" -o t -- $T64 > /dev/null 2> "$OUT/tree64_pmc.err"
# 7. the register-tile kernels at the sizes of the executor's bench lines: kernel stats (d = 16 x 4096 and d = 32 x 2048 replicas, the walk)
for cfg in "16 64 4096" "32 32 2048"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tree$1" -o tree$1 -- python $ROOT/scripts/prof_tree_wave.py $1 $2 $3 3 > "$OUT/driver_tree$1.txt" 2> "$OUT/tree$1.err"
done
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, hashlib, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
def rows(sub):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        yield from csv.DictReader(open(f))
def avg(counter, sub, names, last=5):
    vals = {}
    for r in rows(sub):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].split("(")[0]
        for k in names:
            if k in n:
                vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v[-last:]) / len(v[-last:]) for k, v in vals.items()}
def total(counter, sub, needle):
    """sum of the counter over every dispatch whose kernel name holds `needle`, and the number of such dispatches"""
    s, n = 0.0, 0
    for r in rows(sub):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            s += float(r["Counter_Value"]); n += 1
    return s, n
sha = lambda f: hashlib.sha256(open(os.path.join("rxinfer.jl_amd", "csrc", f), "rb").read()).hexdigest()
c2 = ("k_seg_aggregate", "k_boundary_scan", "k_forward", "k_backward")
fetch, write = avg("FETCH_SIZE", "pmc_fetch", c2), avg("WRITE_SIZE", "pmc_write", c2)
t = {"source": f"profiles/{tag}/ (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, scripts/prof_driver.py C2 workload, shared-model batch)",
     "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced streaming reads (MI355X_MICROARCH.md HBM section; confirmed by scripts/fetch_calib.hip, fetch_calibration.txt); KiB -> bytes x1024"}
for k in ("k_seg_aggregate", "k_forward", "k_backward"):
    if k in fetch and k in write:
        t[f"{k}_fetch_bytes_per_launch"] = fetch[k] * 1024 * 2
        t[f"{k}_write_bytes_per_launch"] = write[k] * 1024
        t[f"{k}_hbm_bytes_per_launch"] = fetch[k] * 1024 * 2 + write[k] * 1024
t["lgssm_kernels_sha256"] = sha("lgssm_kernels.hpp")
json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
kern = ("kd_forward_info", "kd_backward_info", "kd_fe_resid_mfma", "kd_agg_gemm", "kd_scan_fix", "kd_scan_local")
m = {"source": f"profiles/{tag}/ (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 …, scripts/prof_driver.py --config c3: d = dy = 64, T = 10^4, one chain; average per dispatch over the timed sweeps)",
     "flop_per_instruction": 2048, "mfma_f64_per_launch": avg("SQ_INSTS_VALU_MFMA_F64", "c3pmc_a", kern, last=3),
     "mfma_busy_cycles_per_launch": avg("SQ_VALU_MFMA_BUSY_CYCLES", "c3pmc_c", kern, last=3), "busy_cycles_per_launch": avg("SQ_BUSY_CYCLES", "c3pmc_a", kern, last=3),
     "dense_kernels_sha256": sha("dense_kernels.hpp")}
json.dump(m, open(os.path.join(out, "mfma_insts.json"), "w"), indent=1)
v = {"source": f"profiles/{tag}/ (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES, scripts/prof_driver.py --config c4 / c5; per launch = per filtering pass of 4096 series x 2000 observations x 10 iterations (c4), per VMP iteration over 1e7 points (c5))"}
for cfg, kname in (("c4", "k_hgf_filter"), ("c5", "k_gmm_pass")):
    d = {"kernel": kname}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVES"):
        a = avg(c, cfg + "pmc", (kname,), last=3)
        if kname in a:
            d[c] = a[kname]
    v[cfg] = d
v["hgf_kernels_sha256"], v["gmm_kernels_sha256"] = sha("hgf_kernels.hpp"), sha("gmm_kernels.hpp")
json.dump(v, open(os.path.join(out, "valu_insts.json"), "w"), indent=1)
# FETCH_SIZE / WRITE_SIZE calibration: true bytes per launch 4 GiB
true = float(4 << 30)
cal = {}
for k in ("k_calib_read8", "k_calib_read16", "k_calib_slots8"):
    a = avg("FETCH_SIZE", "calib_fetch", (k,), last=3)
    if k in a:
        cal[k] = {"FETCH_SIZE_KiB": a[k], "true_bytes": true, "factor": true / (a[k] * 1024)}
a = avg("WRITE_SIZE", "calib_write", ("k_calib_write8",), last=3)
if "k_calib_write8" in a:
    cal["k_calib_write8"] = {"WRITE_SIZE_KiB": a["k_calib_write8"], "true_bytes": true, "factor": true / (a["k_calib_write8"] * 1024)}
with open(os.path.join(out, "fetch_calibration.txt"), "w") as f:
    f.write("# scripts/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: 4 GiB through each access pattern; factor = true bytes / (counter x 1024)\n")
    f.write(json.dumps(cal, indent=1) + "\n")
f8 = cal.get("k_calib_slots8", cal.get("k_calib_read8", {})).get("factor")
# the node-array executor (d = 4): all of an iteration's launches — strand levels, Bethe levels, the sum — over the iterations run
drv = None
try:
    drv = eval([l for l in open(os.path.join(out, "driver_tree.txt")) if l.startswith("{")][-1])
except Exception:
    pass
iters = 4   # prof_tree.py: one warm-up run + 3 timed
tf, nf = total("FETCH_SIZE", "tree_fetch", "k_tree_")
tw, nw = total("WRITE_SIZE", "tree_write", "k_tree_")
sf, _ = total("FETCH_SIZE", "tree_fetch", "k_tree_scatter")
sw, _ = total("WRITE_SIZE", "tree_write", "k_tree_scatter")
tt = {"source": f"profiles/{tag}/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, scripts/prof_tree.py 128 65536 3: the bench's node_array workload — two observation branches per state, d = 4, T = 128, 65 536 replicas, strand schedule; the sum over ALL k_tree_* dispatches of an iteration (strand levels, Bethe levels, free-energy sums), averaged over the {iters} iterations run, set_data's scatter left out)",
      "fetch_factor_8B_per_lane": f8, "fetch_factor_note": "true bytes / (FETCH_SIZE x 1024) of a 4 GiB walk in the executor's own access pattern (scripts/fetch_calib.hip k_calib_slots8, fetch_calibration.txt)",
      "moved_bytes_per_sweep": (drv["info"]["bytes_per_sweep"] + drv["info"]["fe_bytes_per_sweep"]) * drv["replicas"] if drv else None,
      "io_bytes_per_sweep": drv["info"]["io_bytes_per_sweep"] * drv["replicas"] if drv else None,
      "tree_kernels_sha256": sha("tree_kernels.hpp"), "dispatches_per_iteration": (nf // iters) if nf else None,
      "fetch_KiB_per_iteration": (tf - sf) / iters, "write_KiB_per_iteration": (tw - sw) / iters}
if f8:
    tt["hbm_bytes_per_iteration"] = (tf - sf) / iters * 1024 * f8 + (tw - sw) / iters * 1024 * cal.get("k_calib_write8", {}).get("factor", 1.0)
json.dump(tt, open(os.path.join(out, "tree_traffic.json"), "w"), indent=1)
# the executor at d = 64: MFMA instructions executed per iteration (all k_wave_* dispatches / iterations run)
d64 = None
try:
    d64 = eval([l for l in open(os.path.join(out, "driver_tree64.txt")) if l.startswith("{")][-1])
except Exception:
    pass
mf, nmf = total("SQ_INSTS_VALU_MFMA_F64", "tree64_pmc", "k_wave_")
vi, _ = total("SQ_INSTS_VALU", "tree64_pmc", "k_wave_")
it64 = d64["iterations_run"] if d64 else 4
tm = {"source": f"profiles/{tag}/ (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 …, scripts/prof_tree_wave.py 64 16 256 3: two observation branches per state, d = 64, dy = 64 + 32, T = 16, 256 replicas; the sum over all k_wave_* dispatches, per iteration)",
      "flop_per_instruction": 2048, "mfma_f64_per_iteration": mf / it64 if nmf else None, "valu_insts_per_iteration": vi / it64 if nmf else None,
      "workload": {"d": 64, "T": 16, "replicas": 256}, "tree_wave_kernels_sha256": sha("tree_wave_kernels.hpp")}
json.dump(tm, open(os.path.join(out, "tree_mfma.json"), "w"), indent=1)
for x in (t, m, v, cal, tt, tm):
    print(json.dumps(x, indent=1))
PY
cat "$OUT/summary.txt" | cut -c1-400
cat "$OUT/driver_tree.txt" "$OUT/driver_tree64.txt" "$OUT/driver_c3.txt" | grep -v 'RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
for d in bench c3 tree tree64 tree16 tree32; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$d.csv"; done
python3 scripts/trace_tree_launches.py "$OUT/tree" 8 > "$OUT/tree_strand_levels.txt" 2>&1
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*_agent_info.csv" -delete
rm -f "$OUT/fetch_calib"
