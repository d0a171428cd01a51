#!/usr/bin/env python3
"""Torch-free driver for rocprofv3: the node-array executor above d = 8 (LDS-staged kernels, csrc/tree_wave_kernels.hpp) on the bench's two-branch chain:
prof_tree_wave.py d T replicas [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
import numpy as np  # noqa: E402

from rxhip import workloads  # noqa: E402
from rxhip.graph import two_branch_chain_graph  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

d, T, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
m = workloads.random_model(d, d, seed=100 * d + d)
h = max(1, d // 2)
gb, xs, ys = two_branch_chain_graph(T, m["A"], m["B"], m["B"][:h], m["P"], m["Q"], m["Q"][:h, :h], m["m0"], m["V0"])
rows = np.random.default_rng(0).standard_normal((R, T * (d + h))) * 2.0
with TreeEngine(gb, n_replicas=R) as eng:
    eng.set_data(ys, rows)
    eng.run(1, True)
    dev = []
    for _ in range(steps):
        eng.run(1, True)
        dev.append(eng.last_iteration_ms())
    print({"config": "node_array two_branch wave", "d": d, "T": T, "replicas": R, "iterations_run": steps + 1, "device_ms_per_step": min(dev), "info": eng.info,
           "rule_calls_per_s": eng.counters()["rule_calls"] / (min(dev) * 1e-3)})
