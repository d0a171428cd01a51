"""Factor graphs OUTSIDE the pattern-matched families (what `rxhip_create` used to answer with RXHIP_ERR_UNSUPPORTED) for the tests of the
level-scheduled executor, and a brute-force checker: the joint Gaussian of all random variables conditioned on the data.

Every builder returns (GraphBuilder, data variable ids, dict of named variable lists)."""
import numpy as np

from rxhip import _lib
from rxhip.graph import GraphBuilder


def _spd(rng, d, scale=1.0):
    a = rng.standard_normal((d, d))
    return scale * (a @ a.T / d + 0.5 * np.eye(d))


def two_branch_chain(T, d=3, dy1=2, dy2=1, seed=0, precision_spelling=False):
    """x[t] ~ N(A x[t-1], P) with TWO observation branches per state: y1[t] ~ N(B1 x[t], Q1), y2[t] ~ N(B2 x[t], Q2)
    (graph_lowering.hpp: "a state with two observation branches")"""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = q @ np.diag(rng.uniform(0.5, 0.95, d)) @ q.T
    B1, B2 = rng.standard_normal((dy1, d)), rng.standard_normal((dy2, d))
    P, Q1, Q2, V0 = _spd(rng, d, 0.1), _spd(rng, dy1), _spd(rng, dy2, 2.0), _spd(rng, d, 4.0)
    gb = GraphBuilder()
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(rng.standard_normal(d)), gb.constvar(V0))
    xs, ys = [], []
    for t in range(T):
        if t:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A), x)
            xn = gb.randomvar(d)
            if precision_spelling:
                gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, xn, a, gb.constvar(np.linalg.inv(P)))
            else:
                gb.mvnormal_mean_cov(xn, a, gb.constvar(P))
            x = xn
        for B, Q in ((B1, Q1), (B2, Q2)):
            b = gb.randomvar(B.shape[0])
            gb.multiply(b, gb.constvar(B), x)
            y = gb.datavar(B.shape[0])
            gb.mvnormal_mean_cov(y, b, gb.constvar(Q))
            ys.append(y)
        xs.append(x)
    return gb, ys, dict(x=xs)


def branching_tree(depth=3, fanout=2, d=2, seed=1, observe_leaves_only=False):
    """A root state with `fanout` children per node down to `depth`: x_child ~ N(A_k x_parent + c_k, P_k); nodes are observed through their own
    maps, some through `+` with a second random root (u ~ N(m_u, V_u); z = B x + u; y ~ N(z, Q)).  A tree that is not a chain."""
    rng = np.random.default_rng(seed)
    gb = GraphBuilder()
    root = gb.randomvar(d)
    gb.mvnormal_mean_cov(root, gb.constvar(rng.standard_normal(d)), gb.constvar(_spd(rng, d, 3.0)))
    xs, ys, us = [root], [], []
    frontier = [(root, 0)]
    while frontier:
        x, lvl = frontier.pop(0)
        leaf = lvl == depth
        if leaf or not observe_leaves_only:
            dy = 1 + (len(ys) % d)
            b = gb.randomvar(dy)
            gb.multiply(b, gb.constvar(rng.standard_normal((dy, d))), x)
            if len(ys) % 3 == 1:   # an additive random disturbance with its own prior
                u = gb.randomvar(dy)
                gb.mvnormal_mean_cov(u, gb.constvar(rng.standard_normal(dy)), gb.constvar(_spd(rng, dy, 0.3)))
                z = gb.randomvar(dy)
                gb.node(_lib.NODE_ADD, z, b, u)
                us.append(u)
                b = z
            y = gb.datavar(dy)
            gb.mvnormal_mean_cov(y, b, gb.constvar(_spd(rng, dy, 0.5)))
            ys.append(y)
        if not leaf:
            for k in range(fanout):
                a = gb.randomvar(d)
                gb.multiply(a, gb.constvar(0.8 * rng.standard_normal((d, d))), x)
                w = gb.randomvar(d)
                gb.node(_lib.NODE_ADD, w, gb.constvar(rng.standard_normal(d)), a) if k % 2 else gb.node(_lib.NODE_ADD, w, a, gb.constvar(rng.standard_normal(d)))
                c = gb.randomvar(d)
                gb.mvnormal_mean_cov(c, w, gb.constvar(_spd(rng, d, 0.2)))
                xs.append(c)
                frontier.append((c, lvl + 1))
    return gb, ys, dict(x=xs, u=us)


def scalar_tree(n_leaves=5, seed=2):
    """scalars: m ~ N(0, 10); leaf_k ~ N(mean = a_k m, var = v_k) through `*`; y_k ~ N(mean = leaf_k + c_k, precision = τ_k) — the
    NormalMeanVariance / NormalMeanPrecision spellings with constant parameters on a star"""
    rng = np.random.default_rng(seed)
    gb = GraphBuilder()
    m = gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, m, gb.constvar(0.3), gb.constvar(10.0))
    ys, leaves = [], []
    for k in range(n_leaves):
        a = gb.randomvar(1)
        gb.multiply(a, gb.constvar(float(rng.uniform(0.5, 2.0))), m)
        lf = gb.randomvar(1)
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, lf, a, gb.constvar(float(rng.uniform(0.1, 1.0))))
        s = gb.randomvar(1)
        gb.node(_lib.NODE_ADD, s, lf, gb.constvar(float(rng.standard_normal())))
        y = gb.datavar(1)
        gb.node(_lib.NODE_NORMAL_MEAN_PRECISION, y, s, gb.constvar(float(rng.uniform(0.5, 3.0))))
        ys.append(y)
        leaves.append(lf)
    return gb, ys, dict(m=[m], leaf=leaves)


def star(n_leaves=300, d=1, seed=6):
    """a hub: m ~ N(m0, V0); y_i ~ N(B_i m, Q_i), i = 1 … n — the mean of n observations with constant noise (a variable of degree n + 1: its marginal
    and every product toward a neighbour are TREES of partial products in the executor)"""
    rng = np.random.default_rng(seed)
    gb = GraphBuilder()
    m = gb.randomvar(d)
    gb.mvnormal_mean_cov(m, gb.constvar(rng.standard_normal(d)), gb.constvar(_spd(rng, d, 5.0)))
    ys = []
    for i in range(n_leaves):
        if i % 3 == 0:   # a leaf through a map: the hub's message toward `*` is a product of all the others
            b = gb.randomvar(d)
            gb.multiply(b, gb.constvar(np.eye(d) + 0.1 * rng.standard_normal((d, d))), m)
        else:
            b = m
        y = gb.datavar(d)
        gb.mvnormal_mean_cov(y, b, gb.constvar(_spd(rng, d, 1.0 + (i % 5))))
        ys.append(y)
    return gb, ys, dict(m=[m])


def chain_with_prediction(T, H, d=2, dy=2, seed=3):
    """a chain whose last H observation variables are RANDOM (no data): their marginals are the predictions"""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A, B = q * 0.9, rng.standard_normal((dy, d))
    P, Q = _spd(rng, d, 0.1), _spd(rng, dy)
    gb = GraphBuilder()
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(np.zeros(d)), gb.constvar(4.0 * np.eye(d)))
    xs, ys, preds = [], [], []
    for t in range(T + H):
        if t:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A), x)
            xn = gb.randomvar(d)
            gb.mvnormal_mean_cov(xn, a, gb.constvar(P))
            x = xn
        b = gb.randomvar(dy)
        gb.multiply(b, gb.constvar(B), x)
        y = gb.datavar(dy) if t < T else gb.randomvar(dy)
        gb.mvnormal_mean_cov(y, b, gb.constvar(Q))
        (ys if t < T else preds).append(y)
        xs.append(x)
    return gb, ys, dict(x=xs, pred=preds)


def chain_state_noise_precision(T, d=2, dy=2, seed=4, also_obs_noise=False, gamma=False):
    """x[t] ~ MvNormal(μ = A x[t-1], Λ = W), W ~ Wishart(ν, S) — the UNKNOWN STATE-noise precision (graph_lowering.hpp rejects it); optionally an unknown
    observation-noise precision R as well.  d = dy = 1 with gamma=True: the Normal / Gamma spelling."""
    rng = np.random.default_rng(seed)
    if gamma:
        d = dy = 1
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A, B = q * 0.9, (rng.standard_normal((dy, d)) if not gamma else np.array([[1.3]]))
    Q = _spd(rng, dy)
    gb = GraphBuilder()
    W = gb.randomvar(d, name="W")
    if gamma:
        gb.node(_lib.NODE_GAMMA_SHAPE_RATE, W, gb.constvar(2.0), gb.constvar(0.5))
        gb.initialize(W, _lib.INIT_GAMMA, [2.0, 1.0])
    else:
        gb.node(_lib.NODE_WISHART, W, gb.constvar(float(d + 2)), gb.constvar(np.eye(d) * 2.0))
        gb.initialize(W, _lib.INIT_WISHART, np.concatenate([[d + 2.0], np.eye(d).ravel()]))
    R = None
    if also_obs_noise:
        R = gb.randomvar(dy, name="R")
        gb.node(_lib.NODE_WISHART, R, gb.constvar(float(dy + 1)), gb.constvar(np.eye(dy)))
        gb.initialize(R, _lib.INIT_WISHART, np.concatenate([[dy + 1.0], np.eye(dy).ravel()]))
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(np.zeros(d)), gb.constvar(4.0 * np.eye(d))) if not gamma else \
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, gb.constvar(0.0), gb.constvar(4.0))
    xs, ys = [], []
    for t in range(T):
        if t:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A if not gamma else float(A[0, 0])), x)
            xn = gb.randomvar(d)
            gb.node(_lib.NODE_NORMAL_MEAN_PRECISION if gamma else _lib.NODE_MVNORMAL_MEAN_PRECISION, xn, a, W)
            x = xn
        b = gb.randomvar(dy)
        gb.multiply(b, gb.constvar(B if not gamma else float(B[0, 0])), x)
        y = gb.datavar(dy)
        if R is not None:
            gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, y, b, R)
        elif gamma:
            gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, b, gb.constvar(float(Q[0, 0])))
        else:
            gb.mvnormal_mean_cov(y, b, gb.constvar(Q))
        xs.append(x); ys.append(y)
    return gb, ys, dict(x=xs, W=[W] + ([R] if R is not None else []))


def known_mean_precision(n, d, seed=7, gamma=False):
    """`mv_iid_wishart_known_mean` (test/models/iid/mv_iid_precision_known_mean_tests.jl:11-16): P ~ Wishart(d + 1, I); y[i] ~ MvNormal(μ = m, Λ = P) with a
    CONSTANT mean — a graph without a single Gaussian random variable.  gamma: the scalar spelling τ ~ Gamma(2, 0.5), y[i] ~ Normal(mean = m, precision = τ).
    Returns (builder, data variables, dict(W=[P], m=mean, prior=(ν0, S0)))."""
    rng = np.random.default_rng(seed)
    if gamma:
        d = 1
    gb = GraphBuilder()
    W = gb.randomvar(d, name="P")
    if gamma:
        gb.node(_lib.NODE_GAMMA_SHAPE_RATE, W, gb.constvar(2.0), gb.constvar(0.5))
        prior = (4.0, np.array([[1.0]]))          # Gamma(a, b) = Wishart_1(2a, 1/(2b))
    else:
        gb.node(_lib.NODE_WISHART, W, gb.constvar(float(d + 1)), gb.constvar(np.eye(d)))
        prior = (float(d + 1), np.eye(d))
    m = rng.random(d)
    mc = gb.constvar(m if d > 1 else float(m[0]))
    ys = []
    for _ in range(n):
        y = gb.datavar(d)
        gb.node(_lib.NODE_NORMAL_MEAN_PRECISION if gamma else _lib.NODE_MVNORMAL_MEAN_PRECISION, y, mc, W)
        ys.append(y)
    return gb, ys, dict(W=[W], m=m, prior=prior)


def known_mean_closed_form(y, m, nu0, S0):
    """the conjugate answer of that model: q(P) = Wishart(ν0 + n, (S0⁻¹ + Σ (y − m)(y − m)')⁻¹) after ONE update, and the Bethe free energy = −log evidence
    = ½ n d log 2π − log Z(ν_n, V_n) + log Z(ν0, S0) with log Z(ν, V) = ½ ν d log 2 + ½ ν log|V| + log Γ_d(ν / 2) — at every iteration"""
    from scipy.special import multigammaln
    n, d = y.shape
    r = y - m
    Vn = np.linalg.inv(np.linalg.inv(S0) + r.T @ r)
    logz = lambda nu, V: 0.5 * nu * d * np.log(2.0) + 0.5 * nu * np.linalg.slogdet(V)[1] + multigammaln(0.5 * nu, d)
    return nu0 + n, Vn, 0.5 * n * d * np.log(2.0 * np.pi) - (logz(nu0 + n, Vn) - logz(nu0, S0))


def random_forest(seed, n_steps=14, dmax=4, precision_vars=False, det_chains=True, dim_set=(1, 2, 3, 4, 5, 8, 12, 20, 33, 48, 64)):
    """A random acyclic graph of the executor's family, grown one factor group at a time from a root state: noise children (covariance or precision
    spelling, constant or — `precision_vars` — a Wishart / Gamma variable shared by several nodes), children through `*` (rows ≤ columns: the Bethe sum
    stays finite), through `+` with a constant or with a second random root, chains of deterministic nodes (`B (A x) + c`), observations direct or
    through maps, a root whose mean is the sum of two data variables (a derived clamped value), unobserved leaves.  Returns (builder, data variables,
    dict(x = named Gaussian variables, W = precision variables))."""
    rng = np.random.default_rng(50_000 + seed)
    gb = GraphBuilder()
    dims = [d for d in dim_set if d <= dmax] or [1]
    named, ys, precs = [], [], {}

    def noise_node(out, mu, d):
        """out ~ N(mu, ·): constant covariance, constant precision, or a shared precision variable of dimension d"""
        k = rng.integers(0, 4 if precision_vars else 2)
        if k == 0:
            gb.mvnormal_mean_cov(out, mu, gb.constvar(_spd(rng, d, rng.uniform(0.2, 2.0)))) if d > 1 else \
                gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, out, mu, gb.constvar(float(rng.uniform(0.2, 2.0))))
        elif k == 1:
            gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, out, mu, gb.constvar(np.linalg.inv(_spd(rng, d, rng.uniform(0.2, 2.0))))) if d > 1 else \
                gb.node(_lib.NODE_NORMAL_MEAN_PRECISION, out, mu, gb.constvar(float(rng.uniform(0.5, 3.0))))
        else:
            if d not in precs:
                W = gb.randomvar(d, name=f"W{d}")
                if d == 1 and rng.integers(0, 2):
                    gb.node(_lib.NODE_GAMMA_SHAPE_RATE, W, gb.constvar(2.0), gb.constvar(0.5))
                    gb.initialize(W, _lib.INIT_GAMMA, [2.0, 1.0])
                else:
                    gb.node(_lib.NODE_WISHART, W, gb.constvar(float(d + 2)), gb.constvar(np.eye(d) * 2.0))
                    gb.initialize(W, _lib.INIT_WISHART, np.concatenate([[d + 2.0], np.eye(d).ravel()]))
                precs[d] = W
            gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION if d > 1 else _lib.NODE_NORMAL_MEAN_PRECISION, out, mu, precs[d])

    def new_root(d):
        x = gb.randomvar(d)
        if rng.integers(0, 5) == 0:   # the mean is `a + b` of two data variables (test/models/models_tests.jl:242-256)
            a, b, s = gb.datavar(d), gb.datavar(d), gb.randomvar(d)
            gb.node(_lib.NODE_ADD, s, a, b)
            ys.extend([a, b])
            noise_node(x, s, d)
        else:
            gb.mvnormal_mean_cov(x, gb.constvar(rng.standard_normal(d)), gb.constvar(_spd(rng, d, 3.0)))
        named.append(x)
        return x

    def through_map(x, rows=None):
        d = gb.rows[x]
        r = rows if rows is not None else int(rng.integers(1, d + 1))
        a = gb.randomvar(r)
        gb.multiply(a, gb.constvar(rng.standard_normal((r, d)) if (r, d) != (1, 1) else float(rng.uniform(0.5, 2.0))), x)
        return a

    def deterministic(x):
        """x, or x through a short chain of deterministic nodes; returns the variable a noise node may take as its mean"""
        k = rng.integers(0, 6)
        if k == 0:
            return x
        a = through_map(x) if k in (1, 2, 5) else x
        d = gb.rows[a]
        if k in (2, 3):      # + constant (either order)
            w = gb.randomvar(d)
            c = gb.constvar(rng.standard_normal(d))
            gb.node(_lib.NODE_ADD, w, a, c) if rng.integers(0, 2) else gb.node(_lib.NODE_ADD, w, c, a)
            a = w
        elif k == 4:         # + a second random root
            u = gb.randomvar(d)
            gb.mvnormal_mean_cov(u, gb.constvar(rng.standard_normal(d)), gb.constvar(_spd(rng, d, 0.5)))
            named.append(u)
            w = gb.randomvar(d)
            gb.node(_lib.NODE_ADD, w, a, u) if rng.integers(0, 2) else gb.node(_lib.NODE_ADD, w, u, a)
            a = w
        elif k == 5 and det_chains:   # B (A x)
            a = through_map(a)
        return a

    new_root(int(rng.choice(dims)))
    for _ in range(n_steps):
        x = named[int(rng.integers(0, len(named)))]
        mean = deterministic(x)
        d = gb.rows[mean]
        if rng.integers(0, 2):   # an observation
            y = gb.datavar(d)
            noise_node(y, mean, d)
            ys.append(y)
        else:                    # a child state (possibly never observed: a prediction)
            c = gb.randomvar(d)
            noise_node(c, mean, d)
            named.append(c)
    return gb, ys, dict(x=named, W=list(precs.values()))


def random_data(gb, ys, n_replicas, seed=0):
    """[replica][Σ dims of the data variables] in the order of `ys`"""
    rng = np.random.default_rng(1000 + seed)
    return rng.standard_normal((n_replicas, int(sum(gb.rows[v] for v in ys)))) * 1.5


def data_dict(gb, ys, row):
    out, o = {}, 0
    for v in ys:
        out[v] = row[o:o + gb.rows[v]]
        o += gb.rows[v]
    return out


# ------------------------------------------------------------------------------------------------------------------------
def brute_force(gb, data):
    """Exact posterior marginals {var: (mean, cov)} and −log evidence of a graph with constant noise parameters: every random variable
    is written as an affine map of the BASE variables (those no deterministic node defines), the joint density of the base variables is a
    Gaussian in information form, and conditioning is one dense solve."""
    nv = len(gb.kind)
    K_RANDOM, K_DATA, K_CONST = _lib.VARKIND_RANDOM, _lib.VARKIND_DATA, _lib.VARKIND_CONST
    det_out = {}
    for t, ifs in zip(gb.ftype, gb.fiface):
        if t in (_lib.NODE_MULTIPLY, _lib.NODE_ADD):
            det_out[ifs[0]] = (t, ifs)
    base = [v for v in range(nv) if gb.kind[v] == K_RANDOM and v not in det_out]
    off, N = {}, 0
    for v in base:
        off[v] = N
        N += gb.rows[v]
    aff = {}

    def expr(v):
        if v in aff:
            return aff[v]
        r = gb.rows[v]
        if gb.kind[v] == K_CONST:
            e = (np.zeros((r, N)), np.atleast_1d(gb.const_value(v)).astype(float).reshape(r))
        elif gb.kind[v] == K_DATA:
            e = (np.zeros((r, N)), np.asarray(data[v], float).reshape(r))
        elif v in det_out:
            t, ifs = det_out[v]
            if t == _lib.NODE_MULTIPLY:
                A = np.atleast_2d(gb.const_value(ifs[1])).astype(float).reshape(r, gb.rows[ifs[2]])
                C, c = expr(ifs[2])
                e = (A @ C, A @ c)
            else:
                (C1, c1), (C2, c2) = expr(ifs[1]), expr(ifs[2])
                e = (C1 + C2, c1 + c2)
        else:
            C = np.zeros((r, N))
            C[:, off[v]:off[v] + r] = np.eye(r)
            e = (C, np.zeros(r))
        aff[v] = e
        return e

    J, h, const = np.zeros((N, N)), np.zeros(N), 0.0
    for t, ifs in zip(gb.ftype, gb.fiface):
        if t in (_lib.NODE_MULTIPLY, _lib.NODE_ADD):
            continue
        d = gb.rows[ifs[0]]
        M = np.atleast_2d(gb.const_value(ifs[2])).astype(float).reshape(d, d)
        W = np.linalg.inv(M) if t in (_lib.NODE_MVNORMAL_MEAN_COV, _lib.NODE_NORMAL_MEAN_VARIANCE) else M
        (Co, co), (Cm, cm) = expr(ifs[0]), expr(ifs[1])
        D, e = Co - Cm, co - cm
        J += D.T @ W @ D
        h -= D.T @ W @ e
        const += 0.5 * (d * np.log(2 * np.pi) - np.linalg.slogdet(W)[1] + e @ W @ e)
    S = np.linalg.inv(J)
    mu = S @ h
    nle = const - 0.5 * h @ mu - 0.5 * N * np.log(2 * np.pi) + 0.5 * np.linalg.slogdet(J)[1]
    post = {}
    for v in range(nv):
        if gb.kind[v] == K_RANDOM:
            C, c = expr(v)
            post[v] = (C @ mu + c, C @ S @ C.T)
    return post, float(nle)


def mean_field_chain(T, d=2, dy=2, seed=8, branches=1, partial=False):
    """The benchmark chain under `constraints = MeanField()`: every Gaussian node q(out) q(μ) (the deterministic `*` nodes keep their joint, as GraphPPL materialises
    it), `@initialization q(x) = MvNormal(0, 4 I)` for every state; the anonymous `A * x[t-1]` starts as the image of its input's initial marginal.  `branches`:
    observation branches per state; `partial`: only every other transition is mean-field (a graph that mixes structured and factorised nodes).
    Returns (builder, data variables, dict(x=states))."""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = 0.9 * q
    gb = GraphBuilder()
    x = gb.randomvar(d, name="x[1]")
    gb.mvnormal_mean_cov(x, gb.constvar(np.zeros(d)), gb.constvar(4.0 * np.eye(d)))
    xs, ys = [x], []
    for t in range(T):
        if t:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A), x)
            xn = gb.randomvar(d, name=f"x[{t + 1}]")
            gb.mvnormal_mean_cov(xn, a, gb.constvar(_spd(rng, d, 0.3)))
            x = xn
            xs.append(x)
        for _ in range(branches):
            b = gb.randomvar(dy)
            gb.multiply(b, gb.constvar(rng.standard_normal((dy, d))), x)
            y = gb.datavar(dy)
            gb.mvnormal_mean_cov(y, b, gb.constvar(_spd(rng, dy)))
            ys.append(y)
    gb.mean_field()
    if partial:   # every other transition back to the structured q(out, μ)
        n = 0
        for f, t in enumerate(gb.ftype):
            o, mu = gb.fiface[f][0], gb.fiface[f][1]
            if t == _lib.NODE_MVNORMAL_MEAN_COV and gb.kind[o] == gb.kind[mu] == _lib.VARKIND_RANDOM:
                n += 1
                if n % 2 == 0:
                    gb.set_clusters(f, (0, 0, 1))
    for v in xs:
        gb.initialize(v, _lib.INIT_MVNORMAL if d > 1 else _lib.INIT_NORMAL, np.concatenate([np.zeros(d), (4.0 * np.eye(d)).ravel()]))
    return gb, ys, dict(x=xs)


def mean_field_fixed_point(gb, data):
    """The fixed point of Gaussian mean field, in closed form: with the joint density of the BASE variables (those no deterministic node defines) in information
    form (J, h), the base variables fall into clusters — two of them share one when a node that is NOT under q(out) q(μ) touches both — and the optimum is
    q_c = N(μ_c, (J_cc)⁻¹) with μ the EXACT posterior mean.  Returns ({var: (mean, cov)} for every random variable, the variational free energy
    E_q[−log p(x, y)] − Σ_c H[q_c])."""
    nv = len(gb.kind)
    K_RANDOM, K_CONST = _lib.VARKIND_RANDOM, _lib.VARKIND_CONST
    det_out = {ifs[0]: (t, ifs) for t, ifs in zip(gb.ftype, gb.fiface) if t in (_lib.NODE_MULTIPLY, _lib.NODE_ADD)}
    base = [v for v in range(nv) if gb.kind[v] == K_RANDOM and v not in det_out]
    off, N = {}, 0
    for v in base:
        off[v] = N
        N += gb.rows[v]
    aff = {}

    def expr(v):
        if v in aff:
            return aff[v]
        r = gb.rows[v]
        if gb.kind[v] == K_CONST:
            e = (np.zeros((r, N)), np.atleast_1d(gb.const_value(v)).astype(float).reshape(r))
        elif gb.kind[v] != K_RANDOM:
            e = (np.zeros((r, N)), np.asarray(data[v], float).reshape(r))
        elif v in det_out:
            t, ifs = det_out[v]
            if t == _lib.NODE_MULTIPLY:
                A = np.atleast_2d(gb.const_value(ifs[1])).astype(float).reshape(r, gb.rows[ifs[2]])
                C, c = expr(ifs[2])
                e = (A @ C, A @ c)
            else:
                (C1, c1), (C2, c2) = expr(ifs[1]), expr(ifs[2])
                e = (C1 + C2, c1 + c2)
        else:
            C = np.zeros((r, N))
            C[:, off[v]:off[v] + r] = np.eye(r)
            e = (C, np.zeros(r))
        aff[v] = e
        return e

    def support(v):   # base variables a random variable is a function of
        C, _ = expr(v)
        return {b for b in base if np.any(C[:, off[b]:off[b] + gb.rows[b]] != 0.0)}

    parent = {b: b for b in base}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    J, h, const = np.zeros((N, N)), np.zeros(N), 0.0
    for f, (t, ifs) in enumerate(zip(gb.ftype, gb.fiface)):
        if t in (_lib.NODE_MULTIPLY, _lib.NODE_ADD):
            continue
        d = gb.rows[ifs[0]]
        M = np.atleast_2d(gb.const_value(ifs[2])).astype(float).reshape(d, d)
        W = np.linalg.inv(M) if t in (_lib.NODE_MVNORMAL_MEAN_COV, _lib.NODE_NORMAL_MEAN_VARIANCE) else M
        (Co, co), (Cm, cm) = expr(ifs[0]), expr(ifs[1])
        D, e = Co - Cm, co - cm
        J += D.T @ W @ D
        h -= D.T @ W @ e
        const += 0.5 * (d * np.log(2 * np.pi) - np.linalg.slogdet(W)[1] + e @ W @ e)
        cl = gb.clusters_of(f)
        sides = [support(ifs[0]) if gb.kind[ifs[0]] == K_RANDOM else set(), support(ifs[1]) if gb.kind[ifs[1]] == K_RANDOM else set()]
        groups = [sides[0] | sides[1]] if cl[0] == cl[1] else sides   # q(out, μ) joins everything the node touches; q(out) q(μ) each side on its own
        for grp in groups:
            grp = sorted(grp)
            for b in grp[1:]:
                parent[find(b)] = find(grp[0])
    mu = np.linalg.solve(J, h)
    Sq = np.zeros((N, N))
    ent = 0.0
    comps = {}
    for b in base:
        comps.setdefault(find(b), []).append(b)
    for members in comps.values():
        idx = np.concatenate([np.arange(off[b], off[b] + gb.rows[b]) for b in members])
        S = np.linalg.inv(J[np.ix_(idx, idx)])
        Sq[np.ix_(idx, idx)] = S
        ent += 0.5 * (len(idx) * (1.0 + np.log(2 * np.pi)) + np.linalg.slogdet(S)[1])
    energy = const + 0.5 * (mu @ J @ mu + np.trace(J @ Sq)) - h @ mu
    post = {}
    for v in range(nv):
        if gb.kind[v] == K_RANDOM:
            C, c = expr(v)
            post[v] = (C @ mu + c, C @ Sq @ C.T)
    return post, float(energy - ent)


# ------------------------------------------------------------------------------------------------------------------------
def mixture_on_tree(N=10, K=2, d=2, seed=11, latent_out=False, const_switch=False, shared_parent=True, const_precision=False):
    """A mixture layer hanging off a Gaussian tree (test/models/mixtures/gmm_multivariate_tests.jl:6-32 is the flat case):
         μ0 ~ MvNormal;  m[k] ~ MvNormal(mean = A_k * μ0, cov = S_k)   (shared_parent; else m[k] ~ MvNormal(const, S_k))
         w[k] ~ Wishart | Gamma (d = 1) | a constant precision;  s ~ Dirichlet | a constant probability vector;  z[i] ~ Categorical(s)
         x[i] ~ NormalMixture(switch = z[i], m = m, p = w);  latent_out: y[i] ~ MvNormal(mean = B * x[i], cov = Q), else x[i] IS the data
       with the mixture nodes under mean field (the only factorisation with rules there; the builder's default), the Gaussian nodes of the tree structured.
       @initialization marginals on m, w, s (and x)."""
    rng = np.random.default_rng(seed)
    gb = GraphBuilder()
    cent = 4.0 * rng.standard_normal((K, d))
    if shared_parent:
        mu0 = gb.randomvar(d)
        gb.node(_lib.NODE_MVNORMAL_MEAN_COV, mu0, gb.constvar(rng.standard_normal(d)), gb.constvar(_spd(rng, d, 2.0)))
    s = gb.constvar(rng.dirichlet(3.0 * np.ones(K))) if const_switch else gb.randomvar(K)
    if not const_switch:
        gb.node(_lib.NODE_DIRICHLET, s, gb.constvar(1.0 + rng.random(K)))
        gb.initialize(s, _lib.INIT_DIRICHLET, 1.0 + rng.random(K))
    m, w = [], []
    for k in range(K):
        mk = gb.randomvar(d)
        if shared_parent:
            ak = gb.randomvar(d)
            gb.node(_lib.NODE_MULTIPLY, ak, gb.constvar(np.eye(d) + 0.3 * rng.standard_normal((d, d))), mu0)
            gb.node(_lib.NODE_MVNORMAL_MEAN_COV, mk, ak, gb.constvar(_spd(rng, d, 9.0)))
        else:
            gb.node(_lib.NODE_MVNORMAL_MEAN_COV, mk, gb.constvar(cent[k]), gb.constvar(_spd(rng, d, 9.0)))
        gb.initialize(mk, _lib.INIT_MVNORMAL if d > 1 else _lib.INIT_NORMAL, np.concatenate([cent[k], np.ravel(_spd(rng, d, 4.0))]))
        if const_precision:
            wk = gb.constvar(_spd(rng, d, 1.5))
        else:
            wk = gb.randomvar(d)
            if d == 1:
                gb.node(_lib.NODE_GAMMA_SHAPE_RATE, wk, gb.constvar(2.0 + rng.random()), gb.constvar(1.0 + rng.random()))
                gb.initialize(wk, _lib.INIT_GAMMA, (2.0 + rng.random(), 1.0 + rng.random()))
            else:
                gb.node(_lib.NODE_WISHART, wk, gb.constvar(d + 1.0 + rng.random()), gb.constvar(_spd(rng, d, 0.3)))
                gb.initialize(wk, _lib.INIT_WISHART, np.concatenate([[d + 1.5], np.ravel(_spd(rng, d, 0.4))]))
        m.append(mk); w.append(wk)
    ys, xs, zs = [], [], []
    dy = max(1, d - 1) if latent_out else d
    for _ in range(N):
        z = gb.randomvar(1)
        gb.node(_lib.NODE_CATEGORICAL, z, s)
        if latent_out:
            x, bx, y = gb.randomvar(d), gb.randomvar(dy), gb.datavar(dy)
            gb.node(_lib.NODE_NORMAL_MIXTURE, x, z, *m, *w)
            gb.node(_lib.NODE_MULTIPLY, bx, gb.constvar(rng.standard_normal((dy, d))), x)
            gb.node(_lib.NODE_MVNORMAL_MEAN_COV, y, bx, gb.constvar(_spd(rng, dy, 0.5)))
            gb.initialize(x, _lib.INIT_MVNORMAL if d > 1 else _lib.INIT_NORMAL, np.concatenate([rng.standard_normal(d), np.ravel(_spd(rng, d, 3.0))]))
            xs.append(x)
        else:
            y = gb.datavar(d)
            gb.node(_lib.NODE_NORMAL_MIXTURE, y, z, *m, *w)
        ys.append(y); zs.append(z)
    return gb, ys, dict(m=m, W=[] if const_precision else w, x=xs, z=zs, s=s)


def volatility_chain(T=5, seed=13, kappa=0.8, omega=-0.5):
    """A hierarchical Gaussian filter unrolled in time (test/models/statespace/hgf_tests.jl:9-31 is one step of it): a volatility chain z[t] ~ N(z[t−1], σz²) on
    top of a value chain x[t] ~ GCV(x[t−1], z[t], κ, ω) observed through y[t] ~ N(x[t], σy²); q(x[t], x[t−1]) q(z[t]) at the GCV nodes, `@initialization` on every z[t]."""
    rng = np.random.default_rng(seed)
    gb = GraphBuilder()
    z, x = gb.randomvar(1), gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, z, gb.constvar(0.2), gb.constvar(1.5))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, gb.constvar(-0.3), gb.constvar(2.0))
    kv, ov = gb.constvar(kappa), gb.constvar(omega)
    ys, zs, xs = [], [], []
    for t in range(T):
        zn, xn, y = gb.randomvar(1), gb.randomvar(1), gb.datavar(1)
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, zn, z, gb.constvar(0.1 + 0.1 * rng.random()))
        gb.node(_lib.NODE_GCV, xn, x, zn, kv, ov)
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, xn, gb.constvar(0.05 + 0.1 * rng.random()))
        gb.initialize(zn, _lib.INIT_NORMAL, (0.1 * rng.standard_normal(), 1.0 + rng.random()))
        z, x = zn, xn
        ys.append(y); zs.append(zn); xs.append(xn)
    gb.gh_points = 31
    return gb, ys, dict(z=zs, x=xs)
