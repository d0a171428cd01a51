#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (scripts/profile_r04.sh <tag>) -> profiles/r04/: the summary, the kernel statistics, the driver lines, traffic.json /
valu_insts.json (also copied to profiles/) and pmc_c3.txt (the C3 kernel lines of the summary under a header with the derived figures).
    python scripts/collect_profiles_r04.py r04b"""
import json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04b"
src, dst = f"gpurun_out/prof_{tag}", "profiles/r04"
summ = open(f"{src}/summary.txt").read().split("\n")
c3 = [l for l in summ if l.startswith("kd_")]
def ctr(line, k):
    m = re.search(k + r"=([0-9.e+]+)", line)
    return float(m.group(1)) if m else None
def derive(name):
    l = [x for x in c3 if x.startswith(name) and "SQ_" in x][0]
    s = [x for x in c3 if x.startswith(name) and "avg_ns" in x][0]
    mf, ns = ctr(l, "SQ_INSTS_VALU_MFMA_F64"), float(re.search(r"avg_ns=\s*([0-9.]+)", s).group(1))
    wa, wc, valu = ctr(l, "SQ_WAIT_ANY"), ctr(l, "SQ_WAVE_CYCLES"), ctr(l, "SQ_INSTS_VALU")
    lbc, lia = ctr(l, "SQ_LDS_BANK_CONFLICT"), ctr(l, "SQ_LDS_IDX_ACTIVE")
    tf = mf * 2048 / (ns * 1e-9) / 1e12
    return (f"# {name}: {mf:.4g} MFMA per launch = {mf / 1e4:.0f} per time step and workgroup; x 2048 flop / {ns / 1e6:.3f} ms = {tf:.1f} TF/s = {tf / 78.6:.3f} of the fp64 peak "
            f"(under the profiler; bench.py's event time is shorter); SQ_WAIT_ANY {wa / wc:.2f} of the wave cycles; other VALU per MFMA {(valu - mf) / mf:.2f}; "
            f"LDS bank conflicts {lbc / lia:.2f} of the LDS-active cycles")
hdr = [f"# scripts/profile_r04.sh {tag}: the C3 sweep kernels of the shipped library (d = dy = 64, T = 10^4, one chain; S = 1000 x L = 10: four workgroups' worth of segments per CU, two resident) under",
       "# rocprofv3 --pmc (three separate passes) on scripts/prof_driver.py --config c3, averages per dispatch over the timed steps; kd_forward_info with its pivot-tile",
       "# inverses seeded by the previous time step.  SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in units of 4 cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles (= 64 x SQ_INSTS_VALU_MFMA_F64).",
       "# Round-3 kernel for comparison (the first r04 pass): kd_forward_info 7.199e6 MFMA (720 per step), 3.51e7 VALU instructions of which 5.64e6 FMA_F64 (the Cramer solves of the",
       "# exact tile inverse), 0.458 ms under the profiler; seeded, every step full (before the segments learned to leave the matrix work): 7.351e6 = 735 per step, 0.44 ms.",
       "# With repeating segments skipping the products the per-launch MFMA count is whatever the non-repeating steps execute (the figure below is an AVERAGE over all steps).",
       derive("kd_backward_info"), derive("kd_forward_info")]
open(f"{dst}/pmc_c3.txt", "w").write("\n".join(hdr + c3) + "\n")
shutil.copy(f"{src}/summary.txt", f"{dst}/rocprof_summary_r04.txt")
keep = ("masked_small.txt", "c1_breakdown.txt", "notebook_sizes.txt", "split_segments.txt", "create_c3_trace.txt", "c3_clean.txt", "driver_c3.txt", "driver_c4.txt",
        "driver_c5.txt", "driver_masked8.txt", "driver_noise.txt", "bench_under_rocprof.json")
for f in os.listdir(src):
    if f.startswith("kernel_stats_") or f in keep:
        shutil.copy(f"{src}/{f}", f"{dst}/{f}")
for name in ("traffic.json", "valu_insts.json"):
    t = json.load(open(f"{src}/{name}"))
    t["source"] = t["source"].replace(f"profiles/{tag}/", "profiles/r04/")
    json.dump(t, open(f"{dst}/{name}", "w"), indent=1)
    json.dump(t, open(f"profiles/{name}", "w"), indent=1)
print(open(f"{dst}/pmc_c3.txt").read().split("\n")[6][:300])
