#!/usr/bin/env python3
"""Localises a finding of fuzz_executor.py on a random forest: the same graph grown to 1, 2, … factor groups (random_forest's generator is sequential, so a
shorter run is a prefix), the same observations missing, default replica 0; per prefix the worst variable, executor against oracle.  Usage: fuzz_prefix.py <seed>."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("rxinfer.jl_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
os.environ["RXHIP_TEST_HOOKS"] = "1"
import tree_graphs as tg  # noqa: E402
import tree_oracle  # noqa: E402
from fuzz_cases import DIMS  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
assert rng.choice(["forest", "forest", "forest", "mixture", "volatility"]) == "forest"
dmax = int(rng.choice(DIMS))
prec = bool(rng.random() < 0.4)
miss = (not prec) and bool(rng.random() < 0.3)
steps = int(rng.integers(4, 12))
gb, ys, named = tg.random_forest(seed, n_steps=steps, dmax=dmax, precision_vars=prec, dim_set=DIMS)
R = int(rng.choice([1, 2, 3, 70]))
data = tg.random_data(gb, ys, R, seed)
gone = {}
if miss:
    for v in ys:
        for r in range(R):
            if rng.random() < 0.25:
                gone.setdefault(r, set()).add(v)
r = R - 1
print(f"seed {seed}: dmax {dmax}, {steps} steps, R = {R}, precision variables {prec}, missing in replica {r}: {sorted(gone.get(r, ()))}")
for ns in range(0, steps + 1):
    gb, ys, named = tg.random_forest(seed, n_steps=ns, dmax=dmax, precision_vars=prec, dim_set=DIMS)
    y = tg.random_data(gb, ys, 1, seed)
    o = 0
    for v in ys:
        if v in gone.get(r, ()):
            y[0, o:o + gb.rows[v]] = np.nan
        o += gb.rows[v]
    line = f"{ns:3d} steps, {len(gb.ftype)} factors (last {[(int(gb.ftype[f]), tuple(int(i) for i in gb.fiface[f])) for f in range(max(0, len(gb.ftype) - 3), len(gb.ftype))]}): "
    try:
        ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, y[0]), iterations=2 if prec else 1)
    except Exception as e:
        print(line + f"oracle: {e}")
        continue
    try:
        with TreeEngine(gb, n_replicas=1, allow_missing=miss) as eng:
            if ys:
                eng.set_data(ys, y)
            eng.run(2 if prec else 1, True)
            g = tree_oracle.TreeGraph(gb.to_dump())
            gv = [v for v in range(len(gb.kind)) if g.gauss[v]]
            post, fe = eng.marginals(gv), eng.free_energy_per_replica()
        errs = {v: max(float(np.max(np.abs(post[v][0][0] - ref["mean"][v]) / np.sqrt(np.diag(ref["cov"][v])))),
                       float(np.max(np.abs(post[v][1][0] - ref["cov"][v]) / np.outer(np.sqrt(np.diag(ref["cov"][v])), np.sqrt(np.diag(ref["cov"][v])))))) for v in gv}
        w = max(errs, key=errs.get)
        print(line + f"worst var {w} (d = {gb.rows[w]}) {errs[w]:.2e}, fe {fe[0]:.10g} vs {ref['fe'][-1]:.10g}")
    except Exception as e:
        print(line + f"executor: {str(e)[:120]}; oracle fe {ref['fe'][-1]:.6g}, largest variance {max(float(np.max(np.abs(c))) for c in ref['cov'].values()):.3g}")
