"""One case of the randomized differential campaign (scripts/fuzz_executor.py): from a seed, a random graph of the executor's family — a forest over every
construct (tests/tree_graphs.py::random_forest; optionally shared precision variables with VMP iterations, optionally `missing` observations), a mixture layer
or a volatility chain at random sizes — a random replica count, schedule and kernel family; the executor against oracle/tree_oracle.py on one replica.
tests/test_tree_fuzz_gpu.py replays the seeds that found defects."""
import os

import numpy as np
import tree_graphs as tg
import tree_oracle
from rxhip.tree import TreeEngine

STATS = dict(compared=0, refused=0)
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
            # The bar: 1e-7 sd / 1e-8 — loosened where the ORACLE is the limit: it forms every marginal by two inversions, so the image of a state under a nearly
            # singular map (condition c) carries ≈ 10 c·ε there, while the executor pushes the state's marginal through the map (seed 34357: condition 3e9, oracle 3e-7 from
            # brute-force conditioning, executor 8e-16); its entropy terms lose the same digits (seeds 14828, 32341)
            # Where the graph allows (no precision variables, nothing missing) brute-force conditioning of the joint Gaussian is a second reference without that
            # limit: a variable passes if the executor agrees with EITHER at the plain bar.
            conds = {v: float(np.linalg.cond(ref["cov"][v])) for v in gv}
            dense, dense_fe = (None, None) if prec_vars or miss or kind != "forest" else tg.brute_force(gb, tg.data_dict(gb, ys, data[r]))
            for v in gv:
                sd = np.sqrt(np.diag(ref["cov"][v]))
                if np.all(sd < 1e-7):
                    continue
                e = max(float(np.max(np.abs(post[v][0][r] - ref["mean"][v]) / sd)), float(np.max(np.abs(post[v][1][r] - ref["cov"][v]) / np.outer(sd, sd))))
                if dense is not None and v in dense:
                    sb = np.sqrt(np.diag(dense[v][1]))
                    e = min(e, max(float(np.max(np.abs(post[v][0][r] - dense[v][0]) / sb)), float(np.max(np.abs(post[v][1][r] - dense[v][1]) / np.outer(sb, sb))))) / max(1.0, 1e-9 * conds[v])   # (fp64: condition · ε)
                else:
                    e /= max(1.0, 1e-7 * conds[v])
                worst = max(worst, e)
            ef = abs(fe[r] - ref["fe"][-1]) / max(1.0, abs(ref["fe"][-1])) if np.isfinite(ref["fe"][-1]) else 0.0
            if dense_fe is not None and np.isfinite(dense_fe):   # (−log evidence of the joint Gaussian: what the Bethe free energy of a tree is)
                ef = min(ef, abs(fe[r] - dense_fe) / max(1.0, abs(dense_fe)))
            else:
                ef /= max(1.0, 1e-7 * max(conds.values()))
            for w in prec_vars:
                nu, V = eng.precision(w)
                worst = max(worst, max(abs(nu[r] - ref["q_prec"][w][0]) / ref["q_prec"][w][0], float(np.max(np.abs(V[r] - ref["q_prec"][w][1])) / np.max(np.abs(ref["q_prec"][w][1])))) / max(1.0, 1e-7 * max(conds.values())))
            if not (worst < 1e-7 and ef < 1e-8):
                return f"FAIL {tag}: posterior err {worst:.2e}, fe rel {ef:.2e}, condition-scaled where the oracle is the only reference (kernels {eng.info['kernels']}, dmax {eng.info['dmax']}, largest condition {max(conds.values()):.1e})"
    except Exception as e:   # a refusal by name is fine; anything else is a finding
        msg = str(e)
        if "status 2" in msg or "UNSUPPORTED" in msg:
            STATS["refused"] += 1
            return None
        if ("not positive definite" in msg or "not finite" in msg) and miss:   # the dropped observations left a variable without information: improper in the oracle as well?
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
    STATS["compared"] += 1
    return None


def _spd(rng, d, s=1.0):
    a = rng.standard_normal((d, d))
    return s * (a @ a.T / d + 0.5 * np.eye(d))


def run_chain_case(seed, detail=False):
    """One random linear Gaussian state-space chain (tests/test_families_vs_executor_gpu.py's construction at random sizes: state dimension 1 … 64, 2 … 400
    steps, 1 … 70 chains, per-step constants, known and data inputs, `missing` observations) through the pattern-matched engine `rxhip_create` picks and through the
    node-array executor — two implementations that share no kernel.  None if they agree (1e-7 sd, free energy 1e-8) or the lowering refuses by name."""
    from rxhip import graph
    rng = np.random.default_rng(seed)
    d = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 6, 8, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64]))
    dy = int(rng.integers(1, d + 1)) if rng.random() < 0.7 else d
    tmax = 400 if d <= 8 else 100 if d <= 32 else 40
    T = int(np.exp(rng.uniform(np.log(2), np.log(tmax))))
    C = int(rng.choice([1, 2, 3, 17, 70])) if d <= 16 else int(rng.choice([1, 2, 5]))
    ptt = bool(rng.integers(0, 2))
    per_step = rng.random() < 0.3
    nm = 3
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    As = [q @ np.diag(rng.uniform(0.4, 0.95, d)) @ q.T * rng.uniform(0.8, 1.0) for _ in range(nm)]
    Bs = [rng.standard_normal((dy, d)) for _ in range(nm)]
    Ps = [_spd(rng, d, 0.2) for _ in range(nm)]
    Qs = [_spd(rng, dy, 1.0) for _ in range(nm)]
    m0, V0 = rng.standard_normal(d), _spd(rng, d, 3.0)
    pick = rng.integers(0, nm, size=T + 1)
    kw = {}
    if per_step:
        kw.update(A_of_t=lambda t: As[pick[t]], P_of_t=lambda t: Ps[pick[t]], B_of_t=lambda t: Bs[pick[t]], Q_of_t=lambda t: Qs[pick[t]])
    if rng.random() < 0.3:
        cx = rng.standard_normal((T + 1, d))
        kw.update(c_of_t=lambda t: cx[t], const_first=bool(rng.integers(0, 2)))
    if rng.random() < 0.3:
        cy = rng.standard_normal((T + 1, dy))
        kw.update(d_of_t=lambda t: cy[t])
    du, k = 0, rng.random()
    if k < 0.15:
        du = int(rng.integers(1, 3))
        kw.update(Bu=rng.standard_normal((d, du)), du=du)
    elif k < 0.3:
        du = d
        kw.update(du=d)
    pmiss = float(rng.choice([0.0, 0.0, 0.1, 0.4]))
    out = graph.lgssm_graph(T, As[0], Bs[0], Ps[0], Qs[0], m0, V0, prior_through_transition=ptt, **kw)
    gb, xs, ys = out[0], out[1], out[2]
    us = out[3] if du else []
    y = rng.standard_normal((C, T, dy)) * 2.0
    if pmiss:
        y[rng.random((C, T)) < pmiss] = np.nan
    u = rng.standard_normal((C, len(us), du)) if du else None
    mode = int(rng.integers(0, 4))
    os.environ["RXHIP_TREE_MODE"] = str(mode)
    os.environ.pop("RXHIP_TREE_TILE", None)
    tag = f"chain seed {seed}: d={d} dy={dy} T={T} C={C} ptt={ptt} per_step={per_step} c={'c_of_t' in kw} d={'d_of_t' in kw} du={du} missing={pmiss} executor mode={mode}"
    try:
        eng = graph.create_engine_from_graph(gb.tables(n_replicas=C, allow_missing=pmiss > 0)[0])
    except Exception as e:
        if "status 2" in str(e) or "UNSUPPORTED" in str(e):
            STATS["refused"] += 1
            return None
        return f"ERROR {tag}: create: {str(e)[:200]}"
    try:
        eng.set_data(y, layout="chain_time")
        if du:
            un = np.zeros((C, T, du))
            un[:, T - len(us):] = u
            eng.set_inputs(un, layout="chain_time")
        eng.run(1, True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
    except Exception as e:
        return f"ERROR {tag}: engine: {str(e)[:200]}"
    finally:
        eng.close()
    try:
        with TreeEngine(gb, n_replicas=C, allow_missing=pmiss > 0) as te:
            te.set_data(ys, y.reshape(C, T * dy))
            if du:
                te.set_data(us, u.reshape(C, len(us) * du))
            te.run(1, True)
            post = te.marginals(xs)
            tfe = te.free_energy_per_replica()
    except Exception as e:
        # The executor forms q of EVERY variable of the graph, the image B x + c included; behind a square B of condition κ that covariance has condition ≥ κ²: from
        # κ ≈ 1e5 it is singular to fp64 (the reference's cholinv throws on it as well).  The pattern-matched engine never forms it (seeds 202924: κ = 6e5, d = 64).
        if "not positive definite" in str(e) and dy == d and max(np.linalg.cond(b) for b in (Bs if per_step else Bs[:1])) > 1e5:
            STATS["refused"] += 1
            return None
        return f"ERROR {tag}: executor: {str(e)[:200]}"
    tm = np.stack([post[v][0] for v in xs], axis=1)      # [C][T][d]
    tc = np.stack([post[v][1] for v in xs], axis=1)
    sd = np.sqrt(np.einsum("ctii->cti", cov))
    em = float(np.max(np.abs(tm - mean) / sd))
    ec = float(np.max(np.abs(tc - cov) / (sd[..., :, None] * sd[..., None, :])))
    ef = float(np.max(np.abs(tfe - fe) / np.maximum(1.0, np.abs(fe))))
    if not (em < 1e-7 and ec < 1e-7 and ef < 1e-8):
        msg = f"FAIL {tag}: mean {em:.2e} cov {ec:.2e} fe {ef:.2e}"
        if detail:   # the oracle as the arbiter, on the chain with the largest free-energy difference
            c = int(np.argmax(np.abs(tfe - fe) / np.maximum(1.0, np.abs(fe))))
            row = np.concatenate([y[c].ravel()] + ([u[c].ravel()] if du else []))
            try:
                ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, list(ys) + list(us), row))
                sdo = [np.sqrt(np.diag(ref["cov"][v])) for v in xs]
                msg += (f"; chain {c}: free energy engine {fe[c]:.12g}, executor {tfe[c]:.12g}, oracle {ref['fe'][-1]:.12g}; means against the oracle: engine "
                        f"{max(float(np.max(np.abs(mean[c, t] - ref['mean'][v]) / sdo[t])) for t, v in enumerate(xs)):.2e}, executor "
                        f"{max(float(np.max(np.abs(tm[c, t] - ref['mean'][v]) / sdo[t])) for t, v in enumerate(xs)):.2e}")
                g = tree_oracle.TreeGraph(gb.to_dump())
                rest = [v for v in range(len(gb.kind)) if g.gauss[v] and v not in xs]
                with TreeEngine(gb, n_replicas=C, allow_missing=pmiss > 0) as te:
                    te.set_data(ys, y.reshape(C, T * dy))
                    if du:
                        te.set_data(us, u.reshape(C, len(us) * du))
                    te.run(1, True)
                    pr = te.marginals(rest)
                worst = max((float(np.max(np.abs(pr[v][1][c] - ref["cov"][v]) / np.outer(np.sqrt(np.diag(ref["cov"][v])), np.sqrt(np.diag(ref["cov"][v]))))), v, float(np.linalg.cond(ref["cov"][v]))) for v in rest)
                role = {int(gb.fiface[f][0]): (int(gb.ftype[f]), tuple(int(i) for i in gb.fiface[f])) for f in range(len(gb.ftype))}
                errs = sorted(((float(np.max(np.abs(pr[v][1][c] - ref["cov"][v]) / np.outer(np.sqrt(np.diag(ref["cov"][v])), np.sqrt(np.diag(ref["cov"][v]))))), v) for v in rest), reverse=True)
                msg += "; worst five: " + ", ".join(f"{v} {role.get(v)} {e:.1e}" for e, v in errs[:5])
                msg += f"; executor's other variables against the oracle: worst covariance {worst[0]:.1e} (variable {worst[1]}, condition {worst[2]:.1e}); condition of B {[float(f'{np.linalg.cond(b):.2g}') for b in (Bs if per_step else Bs[:1])]}"
            except Exception as e:
                msg += f"; oracle: {str(e)[:120]}"
        return msg
    STATS["compared"] += 1
    return None


def run_engine_case(seed):
    """One random case for the state-space engines' own entry points (`LGSSMEngine`: the path the headline benchmark runs): dimensions 1 … 64, 1 … 600 steps, 1 … 130
    chains, a random number of segments for the parallel-in-time sweep (or the library's choice), either prior placement, `missing` rows, noise scales over four decades
    — smoothing against the oracle's Kalman / RTS restatement (rxo_lgssm_kalman_rts: posteriors, −log evidence), filtering against rxo_lgssm_filter, `filter_step`
    against `run_filter`, on up to three chains of the batch.  None or the finding."""
    import rxhip
    import rxoracle
    from rxhip import workloads
    rng = np.random.default_rng(seed)
    small = rng.random() < 0.6
    d = int(rng.choice([1, 2, 3, 4])) if small else int(rng.choice([5, 6, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64]))
    dy = int(rng.integers(1, d + 1)) if rng.random() < 0.6 else d
    tmax = 600 if d <= 4 else 150 if d <= 16 else 50
    T = int(np.exp(rng.uniform(0.0, np.log(tmax))))
    C = int(rng.choice([1, 2, 3, 64, 70, 128, 130])) if d <= 4 else int(rng.choice([1, 2, 3, 5, 9]))
    segments = 0 if rng.random() < 0.4 else int(rng.integers(1, T + 1))
    ptt = bool(rng.integers(0, 2))
    pmiss = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    mdl = workloads.random_model(d, dy, seed, stable=float(rng.uniform(0.3, 0.99)))
    mdl["P"] = mdl["P"] * 10.0 ** rng.uniform(-2, 1)
    mdl["Q"] = mdl["Q"] * 10.0 ** rng.uniform(-2, 1)
    mdl["V0"] = mdl["V0"] * 10.0 ** rng.uniform(-1, 3)
    y = workloads.generate_batch(mdl, T, C, seed0=seed, threads=1)
    if pmiss:
        y[rng.random((T, C)) < pmiss] = np.nan
    tag = f"engine seed {seed}: d={d} dy={dy} T={T} C={C} segments={segments} ptt={ptt} missing={pmiss}"
    args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    chains = sorted(set(int(c) for c in rng.integers(0, C, size=3)))
    try:
        with rxhip.LGSSMEngine(*args, T=T, n_chains=C, prior_through_transition=ptt, segments=segments, allow_missing=pmiss > 0) as eng:
            eng.set_data(y)
            eng.run(1, True)
            mean, cov = eng.marginals()
            fe = eng.free_energy_per_chain()
            filt = None
            if not pmiss:
                eng.run_filter(free_energy=True)
                fm, fc = eng.marginals()
                ffe = eng.free_energy_per_chain()
                filt = (fm, fc, ffe)
    except Exception as e:
        if "kappa" in str(e) and "status 2" in str(e):   # beyond the conditioning envelope: what a caller gets instead — `infer`'s detour over the node-array executor — held
            try:                                          # to the contract's bars against the same restatement
                spec = rxhip.linear_gaussian_ssm(*args, prior_through_transition=ptt)
                res = rxhip.infer(model=spec, data={"y": np.transpose(y, (1, 0, 2))}, free_energy=True)
                pm, pc, pf = res.posteriors["x"].mean, res.posteriors["x"].cov, res.free_energy
                for c in chains:
                    om, oc, onll = rxoracle.lgssm_kalman_rts(*args, y[:, c], prior_through_transition=ptt)
                    sd = np.sqrt(np.einsum("tii->ti", oc))
                    if not np.all(np.isfinite(sd)):   # (the restatement's own variances went negative: a vaguer prior than its RTS form survives)
                        continue
                    e2 = max(float(np.max(np.abs(pm[c] - om) / sd)), float(np.max(np.abs(pc[c] - oc) / (sd[:, :, None] * sd[:, None, :]))))
                    ef2 = abs(pf[c][0] - onll) / max(1.0, abs(onll))
                    if not (e2 < 1e-6 and ef2 < 1e-8):
                        return f"FAIL {tag}: chain {c} on the executor detour: posterior {e2:.2e} sd, free energy {ef2:.2e}"
                STATS["detour"] = STATS.get("detour", 0) + 1
            except Exception as e3:
                if "not positive definite" not in str(e3) and "status 2" not in str(e3):
                    return f"ERROR {tag}: executor detour: {str(e3)[:200]}"
            STATS["refused"] += 1
            return None
        if "status 2" in str(e) or "UNSUPPORTED" in str(e):
            STATS["refused"] += 1
            return None
        return f"ERROR {tag}: {str(e)[:200]}"
    worst = 0.0
    for c in chains:
        om, oc, onll = rxoracle.lgssm_kalman_rts(*args, y[:, c], prior_through_transition=ptt)
        sd = np.sqrt(np.einsum("tii->ti", oc))
        e = max(float(np.max(np.abs(mean[:, c] - om) / sd)), float(np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :]))))
        ef = abs(fe[c] - onll) / max(1.0, abs(onll))
        if not (e < 1e-6 and ef < 1e-8):   # (the contract's bars: the models' noise scales span four decades)
            return f"FAIL {tag}: chain {c} smoothing: posterior {e:.2e} sd, free energy {ef:.2e} ({fe[c]:.12g} vs {onll:.12g})"
        worst = max(worst, e)
        if filt is not None:
            fm, fc, ffe = filt
            om, oc, ofe, _ = rxoracle.lgssm_filter(*args, y[:, c], ptt)
            sd = np.sqrt(np.einsum("tii->ti", oc))
            e = max(float(np.max(np.abs(fm[:, c] - om) / sd)), float(np.max(np.abs(fc[:, c] - oc) / (sd[:, :, None] * sd[:, None, :]))))
            ef = abs(ffe[c] - ofe) / max(1.0, abs(ofe))
            if not (e < 1e-6 and ef < 1e-8):
                return f"FAIL {tag}: chain {c} filtering: posterior {e:.2e} sd, free energy {ef:.2e} ({ffe[c]:.12g} vs {ofe:.12g})"
    STATS["compared"] += 1
    return None


def run_vmp_case(seed):
    """One random case for the variational engines (SURVEY §8 a9 – a11): the univariate mixture (`GMMEngine`), the multivariate mixture (`MvGMMEngine`) or the
    hierarchical Gaussian filter (`HGFEngine`) at random sizes, priors, initial marginals and iteration counts against the oracle's restatement (rxo_gmm_vmp,
    rxo_mvgmm_vmp, rxo_hgf_filter) — every iteration's posteriors at 1e-6 relative, free energies at 1e-8.  None or the finding."""
    import rxhip
    import rxoracle
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["gmm", "mvgmm", "hgf"]))
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))
    try:
        if kind == "gmm":
            K, n, iters = int(rng.integers(1, 9)), int(np.exp(rng.uniform(0.0, np.log(20000)))), int(rng.integers(1, 13))
            mus = np.sort(rng.uniform(-10 * K, 10 * K, K))
            ws = 10.0 ** rng.uniform(-1, 1.5, K)
            z = rng.choice(K, size=n, p=rng.dirichlet(np.ones(K) * 3))
            y = mus[z] + rng.standard_normal(n) / np.sqrt(ws[z])
            priors = (mus + rng.normal(0, 2, K), np.full(K, 10.0 ** rng.uniform(0, 4)), np.full(K, 10.0 ** rng.uniform(-2, 0.5)), np.full(K, 10.0 ** rng.uniform(-2, 0.5)),
                      np.full(K, float(rng.choice([0.1, 1.0, 5.0]))))
            init = (mus + rng.normal(0, 3, K), np.full(K, 10.0 ** rng.uniform(-1, 3)), np.full(K, 10.0 ** rng.uniform(-1, 1)), np.full(K, 10.0 ** rng.uniform(-3, 1)), np.ones(K))
            tag = f"vmp seed {seed}: gmm K={K} n={n} iterations={iters}"
            with rxhip.GMMEngine(n, *priors, *init) as eng:
                eng.set_data(y)
                eng.run(iters, True)
                hist, fe = eng.history(), eng.free_energy()
            ohist, ofe, _, _ = rxoracle.gmm_vmp(y, *priors, *init, iters)
            e, ef = float(np.max(np.abs(hist - ohist) / np.maximum(np.abs(ohist), 1e-300))), rel(fe, ofe)
        elif kind == "mvgmm":
            d = int(rng.integers(1, 5))
            K = int(rng.integers(1, 17 if d <= 2 else 9))
            N, iters = int(np.exp(rng.uniform(np.log(2.0), np.log(6000)))), int(rng.integers(1, 11))
            means = rng.standard_normal((K, d)) * 10.0 ** rng.uniform(0.5, 1.7)
            y = np.stack([rng.multivariate_normal(means[k], _spd(rng, d, 10.0 ** rng.uniform(-0.5, 1.3))) for k in rng.integers(0, K, N)])
            mu0 = 0.5 * means + rng.uniform(0, 5, (K, d))
            S0 = np.tile(10.0 ** rng.uniform(1, 6) * np.eye(d), (K, 1, 1))
            nu0 = np.full(K, d + float(rng.choice([0.0, 1.0, 4.0])) + 0.001)
            V0 = np.tile(10.0 ** rng.uniform(-1, 2) * np.eye(d), (K, 1, 1))
            al0 = np.full(K, float(rng.choice([0.1, 1.0, 3.0])))
            init = (mu0 + rng.normal(0, 1, (K, d)), np.tile(10.0 ** rng.uniform(0, 4) * np.eye(d), (K, 1, 1)), nu0 + 1.0, V0, np.ones(K))
            tag = f"vmp seed {seed}: mvgmm d={d} K={K} N={N} iterations={iters}"
            with rxhip.MvGMMEngine(N, mu0, S0, nu0, V0, al0, *init) as eng:
                eng.set_data(y)
                eng.run(iters, True)
                h, fe = eng.history(), eng.free_energy()
            ohist, ofe, _ = rxoracle.mvgmm_vmp(y, mu0, S0, nu0, V0, al0, rxoracle.mvgmm_pack(*init), iters)
            o = rxoracle.mvgmm_unpack(ohist, d)
            e, ef = max(rel(h[key], o[key]) for key in ("mean", "cov", "nu", "V", "alpha")), rel(fe, ofe)
        else:
            T, S, iters = int(np.exp(rng.uniform(0.0, np.log(800)))), int(rng.choice([1, 2, 5, 70])), int(rng.integers(1, 13))
            k, w, zv, yv = float(rng.uniform(0.3, 1.5)), float(rng.normal(0, 1)), 10.0 ** rng.uniform(-3, -0.5), 10.0 ** rng.uniform(-3, 0)
            n_gh = int(rng.choice([11, 21, 31]))
            ys = np.empty((T, S))
            for s in range(S):   # the generative model of test/models/statespace/hgf_tests.jl:80-92
                zt, xt = 0.0, 0.0
                for t in range(T):
                    zt = zt + np.sqrt(zv) * rng.standard_normal()
                    xt = xt + np.sqrt(np.exp(k * zt + w)) * rng.standard_normal()
                    ys[t, s] = xt + np.sqrt(yv) * rng.standard_normal()
            tag = f"vmp seed {seed}: hgf T={T} series={S} iterations={iters} kappa={k:.2f} omega={w:.2f} n_gh={n_gh}"
            with rxhip.HGFEngine(T, S, k, w, zv, yv, n_gh=n_gh) as eng:
                eng.set_data(ys)
                eng.run(iters, True)
                zm, zvv, xm, xv = eng.history()
                fe_s = eng.free_energy_per_chain()
            e = ef = 0.0
            for s in sorted(set(int(c) for c in rng.integers(0, S, size=2))):
                o = rxoracle.hgf_filter(ys[:, s], k, w, zv, yv, vmp_iters=iters, n_gh=n_gh)
                e = max(e, rel(zm[:, s], o[0]), rel(zvv[:, s], o[1]), rel(xm[:, s], o[2]), rel(xv[:, s], o[3]))
                efs = abs(fe_s[s] - o[4][-1]) / max(1.0, abs(o[4][-1]))
                if efs >= 1e-8:   # a log-volatility the n_gh-point rule does not reach (an outlier in y pushes z to 9.8 at seed 1409): the integral is not converged in
                    o2 = rxoracle.hgf_filter(ys[:, s], k, w, zv, yv, vmp_iters=iters, n_gh=n_gh + 20)   # n_gh — engine and restatement then differ in how the tails
                    if abs(o2[4][-1] - o[4][-1]) > 1e-4 * max(1.0, abs(o[4][-1])):                         # underflow (posteriors agree to 1e-13); the reference's value is as arbitrary
                        efs = 0.0
                ef = max(ef, efs)
    except Exception as err:
        msg = str(err)
        if "status 2" in msg or "status 3" in msg or "status 4" in msg or "RXO_ERR" in msg or "oracle" in msg.lower():   # refused by name, improper, or a free energy outside the cubature's range — on either side
            STATS["refused"] += 1
            return None
        return f"ERROR vmp seed {seed} ({kind}): {msg[:200]}"
    if not (e < 1e-6 and ef < 1e-8):
        return f"FAIL {tag}: posteriors {e:.2e} relative, free energy {ef:.2e}"
    STATS["compared"] += 1
    return None


def _kalman_rts_numpy(A, B, P, Q, m0, V0, y, ptt, horizon=0):
    """Covariance-form Kalman filter and RTS smoother in numpy with what the oracle's message-order restatements (rxo_lgssm_bp_joints, rxo_lgssm_predict) cannot give
    for a partly observed state (dy < d: their backward message has no covariance): smoothed (mean, cov) of x[1..T+H], the joints q(x[t], A x[t-1]) from the lag-one
    smoother, the messages toward y[1..T+H] (leave-one-out predictive of an observed step: the smoothed belief with the step's own likelihood divided out)."""
    T, d = y.shape[0] + horizon, A.shape[0]
    mp, Vp, mf, Vf = np.empty((T, d)), np.empty((T, d, d)), np.empty((T, d)), np.empty((T, d, d))
    for t in range(T):
        if t == 0:
            mp[0], Vp[0] = (A @ m0, A @ V0 @ A.T + P) if ptt else (m0, V0)
        else:
            mp[t], Vp[t] = A @ mf[t - 1], A @ Vf[t - 1] @ A.T + P
        if t < y.shape[0] and not np.isnan(y[t, 0]):
            S = B @ Vp[t] @ B.T + Q
            K = np.linalg.solve(S, B @ Vp[t]).T
            mf[t] = mp[t] + K @ (y[t] - B @ mp[t])
            IKB = np.eye(d) - K @ B
            Vf[t] = IKB @ Vp[t] @ IKB.T + K @ Q @ K.T   # Joseph form
        else:
            mf[t], Vf[t] = mp[t], Vp[t]
    ms, Vs = mf.copy(), Vf.copy()
    jm, jc = np.empty((T - 1, 2 * d)), np.empty((T - 1, 2 * d, 2 * d))
    for t in range(T - 2, -1, -1):
        J = np.linalg.solve(Vp[t + 1], A @ Vf[t]).T
        ms[t] = mf[t] + J @ (ms[t + 1] - mp[t + 1])
        Vs[t] = Vf[t] + J @ (Vs[t + 1] - Vp[t + 1]) @ J.T
        Vs[t] = 0.5 * (Vs[t] + Vs[t].T)
        lag = Vs[t + 1] @ J.T                      # Cov(x[t+1], x[t])
        jm[t] = np.concatenate([ms[t + 1], A @ ms[t]])
        jc[t] = np.block([[Vs[t + 1], lag @ A.T], [A @ lag.T, A @ Vs[t] @ A.T]])
    dy = B.shape[0]
    pm, pc = np.empty((T, dy)), np.empty((T, dy, dy))
    W = B.T @ np.linalg.solve(Q, B)
    for t in range(T):
        if t < y.shape[0] and not np.isnan(y[t, 0]):
            L = np.linalg.inv(Vs[t])
            Vc = np.linalg.inv(L - W)
            mc = Vc @ (L @ ms[t] - B.T @ np.linalg.solve(Q, y[t]))
        else:
            mc, Vc = ms[t], Vs[t]
        pm[t], pc[t] = B @ mc, B @ Vc @ B.T + Q
    return ms, Vs, jm, jc, pm, pc


def run_option_case(seed):
    """One random case for the OPTIONS of the state-space engines (each a row of the reference's test suite with a restatement in the oracle): known inputs on either
    side (`A x + c[t]`, `B x + d[t]`; per engine or per chain), per-step constants, both, a forecast horizon with predictions, node-local joints, per-chain models,
    the step-wise filter against the whole-series filter, `rxhip_lgssm_infer`, the chain with unknown observation-noise precision.  Model scales stay inside the
    engines' conditioning envelopes (this case is about schedules and options, run_engine_case about conditioning).  None or the finding."""
    import rxhip
    import rxoracle
    from rxhip import workloads
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["offsets", "step", "step+offsets", "horizon", "joints", "chain_model", "filter_step", "infer", "noise", "chain_offsets"]))
    small = kind in ("chain_model", "noise") or rng.random() < 0.5
    d = int(rng.choice([1, 2, 3, 4])) if small else int(rng.choice([5, 6, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64]))
    dy = int(rng.integers(1, d + 1)) if rng.random() < 0.6 else d
    tmax = 400 if d <= 4 else 100 if d <= 16 else 30
    T = int(np.exp(rng.uniform(np.log(2.0), np.log(tmax))))
    C = int(rng.choice([1, 2, 3, 64, 70])) if d <= 4 else int(rng.choice([1, 2, 3]))
    segments = 0 if rng.random() < 0.5 else int(rng.integers(1, T + 1))
    ptt = bool(rng.integers(0, 2))
    M = 3 if kind in ("step", "step+offsets", "chain_model") else 1

    def model(s):
        m = workloads.random_model(d, dy, s, stable=float(rng.uniform(0.4, 0.97)))
        m["P"] = m["P"] * 10.0 ** rng.uniform(-1, 0.5)
        m["Q"] = m["Q"] * 10.0 ** rng.uniform(-1, 0.5)
        m["V0"] = m["V0"] * 10.0 ** rng.uniform(-1, 1.5)
        return m
    ms = [model(seed * 7 + i) for i in range(M)]
    stack = lambda k: np.stack([m[k] for m in ms]) if M > 1 else ms[0][k]
    A, B, P, Q, m0, V0 = (stack(k) for k in ("A", "B", "P", "Q", "m0", "V0"))
    y = workloads.generate_batch(ms[0], T, C, seed0=seed, threads=1) * float(rng.uniform(0.5, 3.0))
    pmiss = float(rng.choice([0.0, 0.0, 0.2])) if kind in ("offsets", "step", "step+offsets", "chain_model", "filter_step") else 0.0
    if pmiss:
        y[rng.random((T, C)) < pmiss] = np.nan
    chains = sorted(set(int(c) for c in rng.integers(0, C, size=3)))
    tag = f"option seed {seed}: {kind} d={d} dy={dy} T={T} C={C} segments={segments} ptt={ptt} missing={pmiss}"
    sdn = lambda oc: np.sqrt(np.einsum("tii->ti", oc))
    perr = lambda mean, cov, om, oc: max(float(np.max(np.abs(mean - om) / sdn(oc))), float(np.max(np.abs(cov - oc) / (sdn(oc)[:, :, None] * sdn(oc)[:, None, :]))))
    ferr = lambda a, b: abs(a - b) / max(1.0, abs(b))
    kw = dict(T=T, n_chains=C, prior_through_transition=ptt, segments=segments, allow_missing=pmiss > 0)
    worst, wfe = 0.0, 0.0
    try:
        if kind in ("offsets", "step", "step+offsets"):
            cx = rng.standard_normal((T, d)) if "offsets" in kind and rng.random() < 0.8 else None
            cy = rng.standard_normal((T, dy)) if "offsets" in kind and (cx is None or rng.random() < 0.6) else None
            sm = rng.integers(0, M, size=T).astype(np.int32) if M > 1 else None
            with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, step_model=sm, state_offset=cx, obs_offset=cy, **kw) as eng:
                eng.set_data(y)
                eng.run(1, True)
                mean, cov = eng.marginals()
                fe = eng.free_energy_per_chain()
            for c in chains:
                if M > 1:
                    om, oc, onll = rxoracle.lgssm_kalman_rts_affine(A, B, P, Q, m0, V0, y[:, c], cx, cy, step_model=sm, prior_through_transition=ptt)
                else:
                    om, oc, onll = rxoracle.lgssm_kalman_rts_affine(A, B, P, Q, m0, V0, y[:, c], cx, cy, prior_through_transition=ptt)
                worst, wfe = max(worst, perr(mean[:, c], cov[:, c], om, oc)), max(wfe, ferr(fe[c], onll))
        elif kind == "chain_offsets":
            cx, cy = rng.standard_normal((T, C, d)), rng.standard_normal((T, C, dy))
            with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, state_offset=np.zeros(d), obs_offset=np.zeros(dy), **kw) as eng:
                eng.set_chain_offsets(cx, cy)
                eng.set_data(y)
                eng.run(1, True)
                mean, cov = eng.marginals()
                fe = eng.free_energy_per_chain()
            for c in chains:
                om, oc, onll = rxoracle.lgssm_kalman_rts_affine(A, B, P, Q, m0, V0, y[:, c], cx[:, c], cy[:, c], prior_through_transition=ptt)
                worst, wfe = max(worst, perr(mean[:, c], cov[:, c], om, oc)), max(wfe, ferr(fe[c], onll))
        elif kind == "horizon":
            H = int(rng.integers(1, 12))
            with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, horizon=H, **kw) as eng:
                eng.set_data(y)
                eng.run(1, True)
                mean, cov = eng.marginals()
                pm, pc = eng.predictions()
            for c in chains:
                if dy == d:   # (the oracle's restatement of the reference's message order; a partly observed state: the numpy smoother above)
                    opm, opc, oxm, oxc = rxoracle.lgssm_predict(A, B, P, Q, m0, V0, y[:, c], horizon=H, prior_through_transition=ptt)
                    om, oc, _ = rxoracle.lgssm_kalman_rts(A, B, P, Q, m0, V0, y[:, c], prior_through_transition=ptt)
                    worst = max(worst, perr(mean[:T, c], cov[:T, c], om, oc), perr(mean[T:, c], cov[T:, c], oxm, oxc), perr(pm[:, c], pc[:, c], opm, opc))
                nm, nc, _, _, npm, npc = _kalman_rts_numpy(A, B, P, Q, m0, V0, y[:, c], ptt, horizon=H)
                worst = max(worst, perr(mean[:, c], cov[:, c], nm, nc), perr(pm[:, c], pc[:, c], npm, npc))
        elif kind == "joints":
            if T < 2:
                return None
            with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, **kw) as eng:
                eng.set_data(y)
                eng.run(1, True)
                jm, jc = eng.node_marginals()
            for c in chains:
                if dy == d:
                    ojm, ojc = rxoracle.lgssm_joints(A, B, P, Q, m0, V0, y[:, c], ptt)
                    worst = max(worst, perr(jm[:, c], jc[:, c], ojm, ojc))
                _, _, njm, njc, _, _ = _kalman_rts_numpy(A, B, P, Q, m0, V0, y[:, c], ptt)
                # (the joint's covariance is singular by construction where A is: compared in units of the marginal standard deviations)
                sdj = np.sqrt(np.maximum(np.einsum("tii->ti", njc), 1e-300))
                worst = max(worst, float(np.max(np.abs(jm[:, c] - njm) / sdj)), float(np.max(np.abs(jc[:, c] - njc) / (sdj[:, :, None] * sdj[:, None, :]))))
        elif kind == "chain_model":
            cm = rng.integers(0, M, size=C).astype(np.int32)
            with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, chain_model=cm, **kw) as eng:
                eng.set_data(y)
                eng.run(1, True)
                mean, cov = eng.marginals()
                fe = eng.free_energy_per_chain()
            for c in chains:
                a = tuple(ms[cm[c]][k] for k in ("A", "B", "P", "Q", "m0", "V0"))
                om, oc, onll = rxoracle.lgssm_kalman_rts(*a, y[:, c], prior_through_transition=ptt)
                worst, wfe = max(worst, perr(mean[:, c], cov[:, c], om, oc)), max(wfe, ferr(fe[c], onll))
        elif kind in ("filter_step", "infer"):
            with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, **kw) as eng:
                if kind == "infer":
                    mean, cov, fe = eng.infer(y, 1, True)
                    fm, fc, ffe = eng.infer(y, 1, True, filtering=True)
                else:
                    fm, fc, ffe = np.empty((T, C, d)), np.empty((T, C, d, d)), np.zeros(C)
                    for t in range(T):
                        fm[t], fc[t], f = eng.filter_step(y[t])
                        ffe += np.where(np.isnan(f), 0.0, f)
                    n_obs = np.maximum(1, np.sum(~np.isnan(y[:, :, 0]), axis=0))
                    ffe = ffe / n_obs
                    mean = None
            for c in chains:
                if mean is not None:
                    om, oc, onll = rxoracle.lgssm_kalman_rts(A, B, P, Q, m0, V0, y[:, c], prior_through_transition=ptt)
                    worst, wfe = max(worst, perr(mean[:, c], cov[:, c], om, oc)), max(wfe, ferr(fe[c], onll))
                if not pmiss:
                    om, oc, ofe, _ = rxoracle.lgssm_filter(A, B, P, Q, m0, V0, y[:, c], ptt)
                    worst, wfe = max(worst, perr(fm[:, c], fc[:, c], om, oc)), max(wfe, ferr(ffe[c], ofe))
                else:   # (missing rows: the filtered belief of the last step is the smoothed one)
                    om, oc, _ = rxoracle.lgssm_kalman_rts(A, B, P, Q, m0, V0, y[:, c], prior_through_transition=ptt)
                    worst = max(worst, perr(fm[-1:, c], fc[-1:, c], om[-1:], oc[-1:]))
        else:   # noise
            iters = int(rng.integers(1, 8))
            nu0, S0 = dy + float(rng.choice([0.5, 2.0, 6.0])), _spd(rng, dy, 10.0 ** rng.uniform(-1, 1))
            with rxhip.LGSSMNoiseEngine(A, B, P, m0, V0, T, nu0, S0, n_chains=C, prior_through_transition=ptt, segments=segments) as eng:
                eng.set_data(y)
                eng.run(iters, True)
                mean, cov = eng.marginals()
                fe = eng.free_energy_per_chain() if C == 1 else None
                nu, V = eng.noise_posterior()
            for c in chains:
                om, oc, wh, ofe = rxoracle.lgssm_noise_vmp(A, B, P, m0, V0, y[:, c], nu0, S0, nu0, S0, iters, ptt)
                worst = max(worst, perr(mean[:, c], cov[:, c], om, oc), abs(nu[c] - wh[-1, 0]) / wh[-1, 0], float(np.max(np.abs(V[c].ravel() - wh[-1, 1:])) / np.max(np.abs(wh[-1, 1:]))))
                if fe is not None:
                    wfe = max(wfe, ferr(fe[0], ofe[-1]))
    except Exception as err:
        msg = str(err)
        if "status 2" in msg or msg.startswith("rxo_"):   # refused by name; or the ORACLE's message-order restatement gives up (rxo_lgssm_bp_joints / rxo_lgssm_predict with dy < d:
            STATS["refused"] += 1                          # the backward message toward a partly observed state has no covariance — where the reference's cholinv throws as well)
            return None
        return f"ERROR {tag}: {msg[:240]}"
    if not (worst < 1e-6 and wfe < 1e-8):
        return f"FAIL {tag}: posteriors {worst:.2e}, free energy {wfe:.2e}"
    STATS["compared"] += 1
    return None
