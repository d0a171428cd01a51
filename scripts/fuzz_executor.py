#!/usr/bin/env python3
"""Differential campaign beyond the test suite's fixed seeds: random forests (tests/tree_graphs.py::random_forest, every construct of the family, optional shared precision
variables and VMP iterations, optional `missing` observations), mixture layers and volatility chains at random sizes — executor (a random schedule and kernel family)
against oracle/tree_oracle.py; with `chains`: random state-space chains through the pattern-matched engines against the executor (tests/fuzz_cases.py run_chain_case); with `engines`: the state-space engines' own entry points (smoothing, filtering; random segment counts, masks, scales) against the oracle's Kalman / RTS restatement (run_engine_case).
Usage: fuzz_executor.py <first seed> <count> [detail | chains [detail] | engines | vmp | options]  (detail: per-variable errors, against brute-force conditioning where the graph allows).  Prints one line per failure and a summary; exit status 1 if anything failed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("rxinfer.jl_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
os.environ["RXHIP_TEST_HOOKS"] = "1"

from fuzz_cases import STATS, run_case, run_chain_case, run_engine_case, run_option_case, run_vmp_case  # noqa: E402

s0, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(s0, s0 + n):
    finding = run_option_case(seed) if sys.argv[3:4] == ["options"] else run_vmp_case(seed) if sys.argv[3:4] == ["vmp"] else run_engine_case(seed) if sys.argv[3:4] == ["engines"] else run_chain_case(seed, detail=len(sys.argv) > 4) if sys.argv[3:4] == ["chains"] else run_case(seed, detail=len(sys.argv) > 3)
    if finding:
        bad += 1
        print(finding, flush=True)
print(f"{n} cases from seed {s0}: {bad} failures ({STATS['compared']} compared, {STATS['refused']} refused by name{', of them ' + str(STATS['detour']) + ' verified on the executor detour' if STATS.get('detour') else ''}, the rest improper)")
sys.exit(1 if bad else 0)
