"""One screen of a bench.py JSON line: the headline, every BASELINE config, first-touch times, parity spots."""
import json, sys
p = json.load(open(sys.argv[1]))
e = p.get("extra") or {}
cb = p.get("cpu_baseline") or {}
print("headline ms", round(p["ms_per_step"], 4), "value", f"{p['value']:.4g}", "frac", round(p["roofline"]["frac"], 4), "traffic_stale", p["roofline"].get("traffic_stale"),
      "parity", (p.get("parity_spot") or {}).get("ok"), "cpu 1-core", f"{cb.get('value', 0):.3g}", "all-cores x", round(cb.get("all_cores", {}).get("value", 0) / max(cb.get("value", 1), 1), 1))
if "roofline_per_chain_models" not in p:   # the compact line bench.py prints (the long form: bench.py --detail <file>): [ms, roofline fraction, parity] per configuration
    for k, v in (p["roofline"].get("per_config") or {}).items():
        print(f"  {k:32s} ms {v[0]:9.4f}  frac {v[1]}  parity {v[2]}")
    sys.exit(0)
r = p["roofline_per_chain_models"]
print("per-chain models ms", round(r["ms_per_step"], 3), "sweep_frac", round(r["sweep_frac"], 3), "| c2_missing ms", round(e["c2_missing"]["ms_per_step"], 3), e["c2_missing"]["parity_spot"]["ok"])
c3 = e["c3"]
print("c1 infer ms", round(e["c1"]["infer_ms"], 4), "(engine built per call:", round(e["c1"].get("infer_ms_engine_built_per_call", 0), 4), ")", "| c3 ms", round(c3["ms_per_step"], 4), c3["kernels_ms_avg"], "first touch", round(c3["create_set_data_first_run_ms"], 2),
      "frac_ref_count", round(c3["frac_ref_count"], 4), "hoisted", round((c3.get("hoisted_matrices") or {}).get("ms_per_step", 0), 4))
print("c4 ms", round(e["c4"]["ms_per_step"], 3), e["c4"]["roofline"]["frac"], "| c5 ms/it", round(e["c5"]["ms_per_iteration"], 4), e["c5"]["roofline"]["frac"])
for k, v in e["mid_sizes"].items():
    print(k, "ms", round(v["ms_per_step"], 4), "first touch", round(v["create_set_data_first_run_ms"], 2), "cov on request", round(v["covariances_on_request"]["ms_per_step"], 4), v["parity_spot"]["ok"])
m = e["masked_mfma"]
print("masked d=64: observed", round(m["fully_observed_ms"], 4), "missing", round(m["missing_10pct_ms"], 4), "stepm", round(m["per_step_constants_4_models_ms"], 4))
n = e.get("lgssm_noise_vmp") or {}
print("noise vmp:", {k: n.get(k) for k in ("ms_per_iteration", "vmp_iters_per_sec", "free_energy_monotone")})
for k in ("c3", "c4", "c5", "lgssm_noise_vmp"):
    print(k, "parity_spot", (e.get(k) or {}).get("parity_spot"))
