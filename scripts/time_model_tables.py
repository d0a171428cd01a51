"""Device time of the once-per-engine table kernels of a shared-model batch (rxhip_get_model_tables_ms)."""
import os
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import rxhip
from rxhip import workloads

mdl = workloads.c1_model()
for T, C in ((100000, 1024), (100000, 64), (10000, 1024), (1000000, 64)):
    os.environ["RXHIP_ONE_PASS"] = "1"
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
        print(f"T={T} chains={C} schedule={eng.schedule()} model_tables_ms={eng.model_tables_ms():.4f}", flush=True)
