#!/usr/bin/env python3
"""Torch-free driver for rocprofv3: the node-array executor on the bench's `node_array` workload (two observation branches per state, d = 4,
T = 256, 4096 replicas; bench.py extra_node_array) through the C ABI only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
import numpy as np  # noqa: E402

from rxhip import workloads  # noqa: E402
from rxhip.graph import two_branch_chain_graph  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

T, R = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), (int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if len(sys.argv) > 4:
    os.environ["RXHIP_TEST_HOOKS"], os.environ["RXHIP_TREE_MODE"] = "1", sys.argv[4]
mdl = workloads.c1_model()
gb, xs, ys = two_branch_chain_graph(T, mdl["A"], mdl["B"][:2], mdl["B"][2:], mdl["P"], mdl["Q"][:2, :2], mdl["Q"][2:, 2:], mdl["m0"], mdl["V0"])
rows = np.random.default_rng(0).standard_normal((R, T * 4)) * 3.0
with TreeEngine(gb, n_replicas=R) as eng:
    eng.set_data(ys, rows)
    eng.run(1, True)
    t0 = time.perf_counter()
    dev = []
    for _ in range(steps):
        eng.run(1, True)
        dev.append(eng.last_iteration_ms())
    dt = (time.perf_counter() - t0) / steps
    info = eng.info
    print({"config": "node_array two_branch", "T": T, "replicas": R, "ms_per_step_wall": dt * 1e3, "device_ms_per_step": min(dev), "info": info,
           "GBps_algorithmic": info["bytes_per_sweep"] * R / (min(dev) * 1e-3) / 1e9, "rule_calls_per_s": eng.counters()["rule_calls"] / (min(dev) * 1e-3)})
