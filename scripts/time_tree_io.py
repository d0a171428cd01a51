"""set_data / get_marginals of the node-array executor at a large batch: host milliseconds."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
from rxhip import workloads  # noqa: E402
from rxhip.graph import two_branch_chain_graph  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

T, R = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mdl = workloads.c1_model()
gb, xs, ys = two_branch_chain_graph(T, mdl["A"], mdl["B"][:2], mdl["B"][2:], mdl["P"], mdl["Q"][:2, :2], mdl["Q"][2:, 2:], mdl["m0"], mdl["V0"])
rows = np.random.default_rng(0).standard_normal((R, T * 4)) * 3.0
with TreeEngine(gb, n_replicas=R) as eng:
    t0 = time.perf_counter()
    eng.set_data(ys, rows)
    t1 = time.perf_counter()
    eng.run(1, True)
    t2 = time.perf_counter()
    post = eng.marginals(xs)
    t3 = time.perf_counter()
    print(f"R={R} T={T}: set_data {1e3 * (t1 - t0):.1f} ms ({rows.nbytes / 1e6:.0f} MB), run {1e3 * (t2 - t1):.1f} ms, marginals of {len(xs)} variables {1e3 * (t3 - t2):.1f} ms "
          f"({sum(post[v][0].nbytes + post[v][1].nbytes for v in xs) / 1e6:.0f} MB)")
