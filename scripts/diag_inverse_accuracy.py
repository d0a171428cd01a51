"""Accuracy of the executor's kernels where a generically ill-conditioned covariance is converted to precision form and back (found by scripts/fuzz_executor.py, seed 3902):
x0 ~ N(m0, V0) with cond(V0) given;  u = A x0 (A random, square);  w = u + c;  y ~ N(w, Λ⁻¹) observed.  Executor against the oracle and against dense conditioning."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("rxinfer.jl_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
os.environ["RXHIP_TEST_HOOKS"] = "1"
import tree_graphs as tg  # noqa: E402
import tree_oracle  # noqa: E402
from rxhip import _lib  # noqa: E402
from rxhip.graph import GraphBuilder  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

rng = np.random.default_rng(0)
for d in (4, 8, 16, 24, 32, 33, 48, 64):
    for cond in (1e2, 1e6, 1e9):
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        V0 = (q * np.logspace(0, np.log10(cond), d)) @ q.T
        V0 = 0.5 * (V0 + V0.T)
        gb = GraphBuilder()
        x0, u, w, y = gb.randomvar(d), gb.randomvar(d), gb.randomvar(d), gb.datavar(d)
        gb.node(_lib.NODE_MVNORMAL_MEAN_COV, x0, gb.constvar(rng.standard_normal(d)), gb.constvar(V0))
        gb.node(_lib.NODE_MULTIPLY, u, gb.constvar(rng.standard_normal((d, d))), x0)
        gb.node(_lib.NODE_ADD, w, u, gb.constvar(rng.standard_normal(d)))
        gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, y, w, gb.constvar(tg._spd(rng, d, 1.0)))
        yv = rng.standard_normal((1, d)) * 3.0
        ref = tree_oracle.infer(gb.to_dump(), {y: yv[0]})
        bf, _ = tg.brute_force(gb, {y: yv[0]})
        with TreeEngine(gb, n_replicas=1) as eng:
            eng.set_data([y], yv)
            eng.run(1, True)
            post = eng.marginals([x0, u, w])
            kern = eng.info["kernels"]
        err = lambda a, b: max(float(np.max(np.abs(a[v][0] - b[v][0]) / np.sqrt(np.diag(b[v][1])))) for v in (x0, u, w))
        dev = {v: (post[v][0][0], post[v][1][0]) for v in (x0, u, w)}
        orc = {v: (ref["mean"][v], ref["cov"][v]) for v in (x0, u, w)}
        print(f"d={d:3d} kernels={kern} cond(V0)={cond:7.0e}   device vs dense conditioning {err(dev, bf):9.2e} sd   oracle vs dense conditioning {err(orc, bf):9.2e} sd", flush=True)
