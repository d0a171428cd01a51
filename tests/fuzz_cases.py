"""One case of the randomized differential campaign (scripts/fuzz_executor.py): from a seed, a random graph of the executor's family — a forest over every
construct (tests/tree_graphs.py::random_forest; optionally shared precision variables with VMP iterations, optionally `missing` observations), a mixture layer
or a volatility chain at random sizes — a random replica count, schedule and kernel family; the executor against oracle/tree_oracle.py on one replica.
tests/test_tree_fuzz_gpu.py replays the seeds that found defects."""
import os

import numpy as np
import tree_graphs as tg
import tree_oracle
from rxhip.tree import TreeEngine

DIMS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 15, 16, 17, 20, 24, 31, 32, 33, 40, 48, 64)


def run_case(seed, detail=False):
    """None if the executor agrees with the oracle (posteriors 1e-7 sd, free energy 1e-8 relative) or refuses the graph by name; the finding as a string otherwise.
    Sets RXHIP_TREE_MODE / RXHIP_TREE_TILE (RXHIP_TEST_HOOKS must be on)."""
    rng = np.random.default_rng(seed)
    kind = rng.choice(["forest", "forest", "forest", "mixture", "volatility"])
    its, miss, prec_vars = 1, False, []
    if kind == "forest":
        dmax = int(rng.choice(DIMS))
        prec = bool(rng.random() < 0.4)
        miss = (not prec) and bool(rng.random() < 0.3)
        its = 2 if prec else 1
        gb, ys, named = tg.random_forest(seed, n_steps=int(rng.integers(4, 12)), dmax=dmax, precision_vars=prec, dim_set=DIMS)
        prec_vars = named["W"]
    elif kind == "mixture":
        its = int(rng.integers(1, 4))
        gb, ys, named = tg.mixture_on_tree(N=int(rng.integers(3, 20)), K=int(rng.integers(1, 5)), d=int(rng.integers(1, 7)), seed=seed, latent_out=bool(rng.random() < 0.4),
                                           const_switch=bool(rng.random() < 0.3), shared_parent=bool(rng.random() < 0.6), const_precision=bool(rng.random() < 0.3))
        prec_vars = named["W"]
    else:
        its = int(rng.integers(1, 4))
        gb, ys, named = tg.volatility_chain(T=int(rng.integers(2, 9)), seed=seed, kappa=0.3 + rng.random(), omega=rng.normal())
    R = int(rng.choice([1, 2, 3, 70]))
    data = tg.random_data(gb, ys, R, seed)
    if miss:
        o = 0
        for v in ys:
            for r in range(R):
                if rng.random() < 0.25:
                    data[r, o:o + gb.rows[v]] = np.nan
            o += gb.rows[v]
    os.environ["RXHIP_TREE_MODE"] = str(int(rng.integers(0, 4)))
    if rng.random() < 0.5:
        os.environ["RXHIP_TREE_TILE"] = str(int(rng.integers(0, 2)))
    else:
        os.environ.pop("RXHIP_TREE_TILE", None)
    tag = f"seed {seed} {kind} R={R} its={its} miss={miss} mode={os.environ['RXHIP_TREE_MODE']} tile={os.environ.get('RXHIP_TREE_TILE')}"
    try:
        with TreeEngine(gb, n_replicas=R, allow_missing=miss) as eng:
            if ys:
                eng.set_data(ys, data)
            eng.run(its, True)
            g = tree_oracle.TreeGraph(gb.to_dump())
            gv = [v for v in range(len(gb.kind)) if g.gauss[v]]
            post, fe = eng.marginals(gv), eng.free_energy_per_replica()
            r = R - 1
            try:
                ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[r]), iterations=its)
            except tree_oracle.ImproperMessage:   # the dropped observations left a message rank-deficient where a rule of the reference wants its covariance (the
                return None                       # reference throws; the executor's precision-form rules may still answer): nothing to compare against
            worst = 0.0
            if detail:   # per variable: executor and oracle against brute-force conditioning of the joint Gaussian (graphs without precision variables)
                print(tag, "factors", [(int(gb.ftype[f]), tuple(int(i) for i in gb.fiface[f])) for f in range(len(gb.ftype))], "data", ys, "missing", np.argwhere(np.isnan(data[r])).ravel())
                bf = None if prec_vars or miss or kind != "forest" else tg.brute_force(gb, {k: a for k, a in tg.data_dict(gb, ys, data[r]).items()})[0]
                for v in gv:
                    sd = np.sqrt(np.diag(ref["cov"][v]))
                    line = f"  var {v} d={gb.rows[v]}: executor vs oracle mean {np.max(np.abs(post[v][0][r] - ref['mean'][v]) / sd):.2e} cov {np.max(np.abs(post[v][1][r] - ref['cov'][v]) / np.outer(sd, sd)):.2e}"
                    if bf is not None and v in bf:
                        sb = np.sqrt(np.diag(bf[v][1]))
                        line += f"; vs brute force: executor {np.max(np.abs(post[v][0][r] - bf[v][0]) / sb):.2e} / {np.max(np.abs(post[v][1][r] - bf[v][1]) / np.outer(sb, sb)):.2e}, oracle {np.max(np.abs(ref['mean'][v] - bf[v][0]) / sb):.2e} / {np.max(np.abs(ref['cov'][v] - bf[v][1]) / np.outer(sb, sb)):.2e}; cond {np.linalg.cond(bf[v][1]):.1e}"
                    print(line)
            for v in gv:
                sd = np.sqrt(np.diag(ref["cov"][v]))
                if np.all(sd < 1e-7):
                    continue
                worst = max(worst, float(np.max(np.abs(post[v][0][r] - ref["mean"][v]) / sd)), float(np.max(np.abs(post[v][1][r] - ref["cov"][v]) / np.outer(sd, sd))))
            ef = abs(fe[r] - ref["fe"][-1]) / max(1.0, abs(ref["fe"][-1])) if np.isfinite(ref["fe"][-1]) else 0.0
            for w in prec_vars:
                nu, V = eng.precision(w)
                worst = max(worst, abs(nu[r] - ref["q_prec"][w][0]) / ref["q_prec"][w][0], float(np.max(np.abs(V[r] - ref["q_prec"][w][1])) / np.max(np.abs(ref["q_prec"][w][1]))))
            if not (worst < 1e-7 and ef < 1e-8):
                return f"FAIL {tag}: posterior err {worst:.2e}, fe rel {ef:.2e} (kernels {eng.info['kernels']}, dmax {eng.info['dmax']})"
    except Exception as e:   # a refusal by name is fine; anything else is a finding
        msg = str(e)
        if "status 2" in msg or "UNSUPPORTED" in msg:
            return None
        if "not positive definite" in msg and miss:   # the dropped observations left a variable without information: improper in the oracle as well?
            try:
                improper = False
                for r in range(R):
                    ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[r]), iterations=its)
                    improper = improper or not np.isfinite(ref["fe"][-1]) or any(np.min(np.linalg.eigvalsh(c)) <= 0 or np.max(np.abs(c)) > 1e12 for c in ref["cov"].values())
            except Exception:
                improper = True
            if improper:
                return None
        return f"ERROR {tag}: {msg[:200]}"
    return None
