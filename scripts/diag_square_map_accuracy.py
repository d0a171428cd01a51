#!/usr/bin/env python3
"""An observation through an ill-conditioned SQUARE map with an offset, `y ~ N(B x + c, Q)`: q(B x + c) is a product of two messages (forward B V Bᵀ in moment form,
backward Q⁻¹ in precision form) — the route on which the chain fuzz (seeds 101839, 119783) found the LDS-staged kernels losing digits.  Per kernel family and
condition number of B: the executor's covariance of w = B x + c and its free energy against the oracle.  Run on an MI355X: python scripts/diag_square_map_accuracy.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("rxinfer.jl_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import tree_graphs as tg  # noqa: E402
import tree_oracle  # noqa: E402
from rxhip import graph  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

for d in (4, 8, 16, 32, 33, 48, 64):
    for cond in (1e2, 1e4, 3e5):
        rng = np.random.default_rng(d)
        U, _ = np.linalg.qr(rng.standard_normal((d, d)))
        W, _ = np.linalg.qr(rng.standard_normal((d, d)))
        B = U @ np.diag(np.geomspace(1.0, cond, d)) @ W.T / np.sqrt(cond)
        spd = lambda s: s * (np.cov(rng.standard_normal((d, 4 * d))) + 0.3 * np.eye(d))
        T = 2
        cy = rng.standard_normal((T + 1, d))
        gb, xs, ys = graph.lgssm_graph(T, 0.9 * np.eye(d), B, spd(0.2), spd(1.0), rng.standard_normal(d), spd(3.0), d_of_t=lambda t: cy[t])[:3]
        y = rng.standard_normal((1, T * d)) * 2.0
        g = tree_oracle.TreeGraph(gb.to_dump())
        gv = [v for v in range(len(gb.kind)) if g.gauss[v]]
        with TreeEngine(gb, n_replicas=1) as te:
            te.set_data(ys, y)
            te.run(1, True)
            post, fe, k = te.marginals(gv), te.free_energy_per_replica()[0], te.info["kernels"]
        ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, y[0]))
        err = {v: float(np.max(np.abs(post[v][1][0] - ref["cov"][v]) / np.outer(np.sqrt(np.diag(ref["cov"][v])), np.sqrt(np.diag(ref["cov"][v]))))) for v in gv}
        w = max(err, key=err.get)
        print(f"d={d:3d} kernels={k} cond(B)={cond:7.0e}: worst covariance {err[w]:.2e} (variable {w}, condition {np.linalg.cond(ref['cov'][w]):.1e}; states {max(err[v] for v in xs):.2e}), "
              f"free energy relative {abs(fe - ref['fe'][-1]) / abs(ref['fe'][-1]):.2e}", flush=True)
