"""Factor-graph tables (include/rxhip.h rxhip_graph_desc) and the lowering entry points.

`GraphBuilder` plays the part of GraphPPL + the inference plugin's walk over the finished model
(src/model/plugins/reactivemp_inference.jl:272-326): variables are created as random / data / constant, factor
nodes as (type, interface → variable).  `lgssm_graph` emits exactly the node sequence RxInfer builds for the
benchmark model: per time step `*`_B, MvNormal_y and (t ≥ 2) `*`_A, MvNormal_x, with one
constant variable per use of A, B, P, Q."""
import ctypes

import numpy as np

from . import _lib
from ._lib import RxHipError


# node vocabulary of include/rxhip.h: code -> (GraphPPL.fform name, interface order); the Julia plugin carries the same table
NODE_VOCABULARY = {
    _lib.NODE_MVNORMAL_MEAN_COV: ("MvNormalMeanCovariance", ("out", "μ", "Σ")),
    _lib.NODE_MULTIPLY: ("*", ("out", "A", "in")),
    _lib.NODE_NORMAL_MEAN_VARIANCE: ("NormalMeanVariance", ("out", "μ", "v")),
    _lib.NODE_NORMAL_MEAN_PRECISION: ("NormalMeanPrecision", ("out", "μ", "τ")),
    _lib.NODE_GAMMA_SHAPE_RATE: ("GammaShapeRate", ("out", "α", "β")),
    _lib.NODE_DIRICHLET: ("Dirichlet", ("out", "a")),
    _lib.NODE_BETA: ("Beta", ("out", "a", "b")),
    _lib.NODE_CATEGORICAL: ("Categorical", ("out", "p")),
    _lib.NODE_BERNOULLI: ("Bernoulli", ("out", "p")),
    _lib.NODE_NORMAL_MIXTURE: ("NormalMixture", ("out", "switch", "m", "p")),
    _lib.NODE_GCV: ("GCV", ("y", "x", "z", "κ", "ω")),
    _lib.NODE_WISHART: ("Wishart", ("out", "ν", "S")),
    _lib.NODE_ADD: ("+", ("out", "in1", "in2")),
    _lib.NODE_MVNORMAL_MEAN_PRECISION: ("MvNormalMeanPrecision", ("out", "μ", "Λ")),
    _lib.NODE_GAMMA_SHAPE_SCALE: ("GammaShapeScale", ("out", "α", "θ")),
}
INIT_FAMILIES = {"normal": _lib.INIT_NORMAL, "gamma": _lib.INIT_GAMMA, "dirichlet": _lib.INIT_DIRICHLET, "mvnormal": _lib.INIT_MVNORMAL,
                 "wishart": _lib.INIT_WISHART}


class GraphBuilder:
    def __init__(self):
        self.kind, self.rows, self.cols, self.coff = [], [], [], []
        self.ftype, self.fiface = [], []
        self.fcluster = {}   # factor index -> cluster id per interface (the node's VariationalConstraintsFactorizationIndicesKey); absent: the schedule's own
        self.pool = []
        self._n = 0
        self.init_family, self.init_off = {}, {}
        self.gh_points = 0
        self.names = []
        self.n_replicas, self.n_observations = 1, 0

    def _var(self, kind, rows, cols=1, coff=-1, name=""):
        self.kind.append(kind); self.rows.append(rows); self.cols.append(cols); self.coff.append(coff)
        self.names.append(name)
        return len(self.kind) - 1

    def randomvar(self, dim, name=""):
        return self._var(_lib.VARKIND_RANDOM, dim, name=name)

    def datavar(self, dim, name=""):
        return self._var(_lib.VARKIND_DATA, dim, name=name)

    def constvar(self, value, name=""):
        v = np.atleast_1d(np.asarray(value, dtype=np.float64))
        rows, cols = (v.shape[0], 1) if v.ndim == 1 else v.shape
        off = self._n
        self.pool.append(v.ravel())
        self._n += v.size
        return self._var(_lib.VARKIND_CONST, rows, cols, off, name=name)

    def const_value(self, var):
        """value of a constant variable (vector, or matrix for cols > 1)"""
        pool = np.concatenate(self.pool)
        v = pool[self.coff[var]:self.coff[var] + self.rows[var] * self.cols[var]]
        return v.reshape(self.rows[var], self.cols[var]) if self.cols[var] > 1 else v.copy()

    # ---- exchange format "rxhip-graph-1": what HIPInferencePlugin.jl's `dump_graph` writes after walking a GraphPPL model ----
    def to_dump(self, n_replicas=1, n_observations=0):
        pool = np.concatenate(self.pool) if self.pool else np.zeros(0)
        kinds = ("random", "data", "constant")
        fams = {v: k for k, v in INIT_FAMILIES.items()}
        ends = sorted([o for o in self.coff if o >= 0] + list(self.init_off.values()) + [pool.size])
        variables = []
        for i in range(len(self.kind)):
            v = {"name": self.names[i], "kind": kinds[self.kind[i]], "rows": int(self.rows[i]), "cols": int(self.cols[i])}
            if self.coff[i] >= 0:
                v["value"] = pool[self.coff[i]:self.coff[i] + self.rows[i] * self.cols[i]].tolist()
            if i in self.init_family:
                o = self.init_off[i]
                v["init"] = {"family": fams[self.init_family[i]], "params": pool[o:min(e for e in ends if e > o)].tolist()}
            variables.append(v)
        factors = []
        for t, ifs in zip(self.ftype, self.fiface):
            name, order = NODE_VOCABULARY[t]
            labels = list(order[:2]) + [f"{order[2]}[{k + 1}]" for k in range((len(ifs) - 2) // 2)] + \
                [f"{order[3]}[{k + 1}]" for k in range((len(ifs) - 2) // 2)] if name == "NormalMixture" else list(order)
            factors.append({"type": name, "interfaces": [[l, int(v)] for l, v in zip(labels, ifs)]})
            if self.fcluster:
                factors[-1]["clusters"] = [int(c) for c in self.clusters_of(len(factors) - 1)]
        return {"format": "rxhip-graph-1", "n_replicas": int(n_replicas), "n_observations": int(n_observations),
                "gh_points": int(self.gh_points), "variables": variables, "factors": factors}

    @classmethod
    def from_dump(cls, dump):
        """Rebuild the tables from a dump (a dict, or the path of a .json / .json.gz file)."""
        if not isinstance(dump, dict):
            import gzip
            import json

            with (gzip.open(dump, "rt") if str(dump).endswith(".gz") else open(dump)) as f:
                dump = json.load(f)
        if dump.get("format") != "rxhip-graph-1":
            raise ValueError("not an rxhip-graph-1 dump")
        codes = {name: code for code, (name, _) in NODE_VOCABULARY.items()}
        gb = cls()
        for v in dump["variables"]:
            if v["kind"] == "constant":
                val = np.asarray(v["value"], dtype=np.float64)
                i = gb.constvar(val.reshape(v["rows"], v["cols"]) if v["cols"] > 1 else val, name=v.get("name", ""))
            else:
                i = (gb.randomvar if v["kind"] == "random" else gb.datavar)(v["rows"], name=v.get("name", ""))
            if "init" in v:
                gb.initialize(i, INIT_FAMILIES[v["init"]["family"]], v["init"]["params"])
        for f in dump["factors"]:
            if f["type"] not in codes:
                raise RxHipError(_lib.ERR_UNSUPPORTED, f"node {f['type']} has no device schedule")
            gb.node(codes[f["type"]], *[int(i) for _, i in f["interfaces"]], clusters=f.get("clusters"))
        gb.gh_points = int(dump.get("gh_points", 0))
        gb.n_replicas, gb.n_observations = int(dump.get("n_replicas", 1)), int(dump.get("n_observations", 0))
        return gb

    def mvnormal_mean_cov(self, out, mu, sigma):
        """out ~ MvNormal(μ = mu, Σ = sigma)  ->  MvNormalMeanCovariance (src/model/graphppl.jl:372-376)"""
        self.ftype.append(_lib.NODE_MVNORMAL_MEAN_COV); self.fiface.append((out, mu, sigma))

    def node(self, ntype, *ifaces, clusters=None):
        """generic factor node: interfaces in the node's declared order; clusters: one id per interface — the factorisation of q around the node
        (GraphPPL.VariationalConstraintsFactorizationIndicesKey, reactivemp_inference.jl:499-506; ((1, 2), (3,)) is 0, 0, 1)"""
        self.ftype.append(ntype); self.fiface.append(tuple(ifaces))
        if clusters is not None:
            self.set_clusters(len(self.ftype) - 1, clusters)

    @staticmethod
    def default_clusters(ntype, n):
        """the factorisation the device schedules implement per node type (include/rxhip.h rxhip_graph_desc.factor_cluster) — what GraphPPL's default
        constraints give the BP families and what the reference tests' @constraints give the VMP ones"""
        if ntype in (_lib.NODE_MVNORMAL_MEAN_COV, _lib.NODE_NORMAL_MEAN_VARIANCE, _lib.NODE_MVNORMAL_MEAN_PRECISION, _lib.NODE_NORMAL_MEAN_PRECISION):
            return (0, 0, 1)
        if ntype in (_lib.NODE_MULTIPLY, _lib.NODE_ADD):
            return (0,) * n
        if ntype == _lib.NODE_GCV:
            return (0, 0) + tuple(range(1, n - 1))
        return tuple(range(n))

    def set_clusters(self, f, clusters):
        if len(clusters) != len(self.fiface[f]):
            raise ValueError("one cluster id per interface")
        self.fcluster[f] = tuple(int(c) for c in clusters)

    def clusters_of(self, f):
        return self.fcluster.get(f, self.default_clusters(self.ftype[f], len(self.fiface[f])))

    def _grouped(self, f, joint):
        """cluster ids as GraphPPL materialises them: clamped (data / constant) interfaces each a cluster of their own, random interfaces joined where
        `joint(k, l)`; clusters numbered by their first interface"""
        ifs = self.fiface[f]
        ids, nxt = [-1] * len(ifs), 0
        for k in range(len(ifs)):
            if ids[k] >= 0:
                continue
            ids[k] = nxt
            if self.kind[ifs[k]] == _lib.VARKIND_RANDOM:
                for l in range(k + 1, len(ifs)):
                    if ids[l] < 0 and self.kind[ifs[l]] == _lib.VARKIND_RANDOM and joint(k, l):
                        ids[l] = nxt
            nxt += 1
        return tuple(ids)

    def bethe(self):
        """GraphPPL's default constraints: all random interfaces of a node in one factor of q (the BP families)"""
        for f in range(len(self.ftype)):
            self.fcluster[f] = self._grouped(f, lambda k, l: True)
        return self

    def gaussian_joint(self):
        """`q(x, W) = q(x)q(W)`: the Gaussian interfaces of a node joint, a random precision interface a factor of its own — the constraints a model with
        Wishart / Gamma precisions needs (under the default the reference has no rule for the joint q(out, μ, W))"""
        gauss = (_lib.NODE_MVNORMAL_MEAN_COV, _lib.NODE_NORMAL_MEAN_VARIANCE, _lib.NODE_MVNORMAL_MEAN_PRECISION, _lib.NODE_NORMAL_MEAN_PRECISION)
        for f, t in enumerate(self.ftype):
            self.fcluster[f] = self._grouped(f, (lambda k, l: k < 2 and l < 2) if t in gauss else (lambda k, l: True))
        return self

    def mean_field(self):
        """`constraints = MeanField()`: every interface of every stochastic node a factor of its own; deterministic nodes (`*`, `+`) keep the joint
        over their random interfaces, as GraphPPL materialises it"""
        for f, t in enumerate(self.ftype):
            det = t in (_lib.NODE_MULTIPLY, _lib.NODE_ADD)
            self.fcluster[f] = self._grouped(f, lambda k, l: det)
        return self

    def initialize(self, var, family, params):
        """`@initialization q(var) = …` (InitMarExtraKey, src/model/plugins/initialization_plugin.jl:201-202)"""
        v = np.atleast_1d(np.asarray(params, dtype=np.float64)).ravel()
        self.init_family[var], self.init_off[var] = family, self._n
        self.pool.append(v)
        self._n += v.size

    def multiply(self, out, A, x):
        """out := A * x  ->  typeof(*) node with an anonymous output variable"""
        self.ftype.append(_lib.NODE_MULTIPLY); self.fiface.append((out, A, x))

    def tables(self, n_replicas=1, permute=None, n_observations=0, allow_missing=False):
        ft = np.asarray(self.ftype, dtype=np.int32)
        order = np.arange(len(ft)) if permute is None else np.asarray(permute)  # node order must not matter to the lowering
        ft = ft[order]
        ifaces = [self.fiface[i] for i in order]
        wide = any(len(t) != 3 for t in ifaces)
        fi = np.asarray([v for t in ifaces for v in t], dtype=np.int64)
        ptr = np.concatenate([[0], np.cumsum([len(t) for t in ifaces])]).astype(np.int64)
        fc = np.asarray([c for i in order for c in self.clusters_of(int(i))], dtype=np.int32) if self.fcluster else None
        nv = len(self.kind)
        fam = np.zeros(nv, dtype=np.int32)
        ioff = np.full(nv, -1, dtype=np.int64)
        for v, f in self.init_family.items():
            fam[v], ioff[v] = f, self.init_off[v]
        arrs = dict(kind=np.asarray(self.kind, dtype=np.int32), rows=np.asarray(self.rows, dtype=np.int32),
                    cols=np.asarray(self.cols, dtype=np.int32), coff=np.asarray(self.coff, dtype=np.int64), ft=np.ascontiguousarray(ft),
                    fi=np.ascontiguousarray(fi), ptr=ptr, fam=fam, ioff=ioff, pool=np.concatenate(self.pool) if self.pool else np.zeros(1))
        g = _lib.GraphDesc()
        g.n_variables = nv
        g.var_kind = arrs["kind"].ctypes.data_as(_lib.c_int32_p)
        g.var_rows = arrs["rows"].ctypes.data_as(_lib.c_int32_p)
        g.var_cols = arrs["cols"].ctypes.data_as(_lib.c_int32_p)
        g.var_const = arrs["coff"].ctypes.data_as(_lib.c_int64_p)
        g.n_factors = len(ft)
        g.factor_type = arrs["ft"].ctypes.data_as(_lib.c_int32_p)
        g.factor_iface = arrs["fi"].ctypes.data_as(_lib.c_int64_p)
        g.const_pool = arrs["pool"].ctypes.data_as(_lib.c_double_p)
        g.n_const = int(arrs["pool"].size if self.pool else 0)
        g.n_replicas = int(n_replicas)
        if wide:  # CSR interface table; 3-wide graphs keep the plain [n_factors][3] form
            g.factor_iface_ptr = arrs["ptr"].ctypes.data_as(_lib.c_int64_p)
        if self.init_family:
            g.var_init_family = arrs["fam"].ctypes.data_as(_lib.c_int32_p)
            g.var_init = arrs["ioff"].ctypes.data_as(_lib.c_int64_p)
        g.gh_points = int(self.gh_points)
        g.n_observations = int(n_observations)
        g.allow_missing = int(bool(allow_missing))
        if fc is not None:
            arrs["fc"] = np.ascontiguousarray(fc)
            g.factor_cluster = arrs["fc"].ctypes.data_as(_lib.c_int32_p)
        g._keep = arrs  # the descriptor points into these arrays
        return g, arrs


def lgssm_graph(T, A, B, P, Q, m0, V0, prior_through_transition=False, A_of_t=None, P_of_t=None, B_of_t=None, Q_of_t=None,
                c_of_t=None, d_of_t=None, const_first=False, Bu=None, du=0):
    # Bu / du: data inputs `A * x[t-1] + B_u * u[t]` (Bu = None with du > 0: u[t] added directly, du = d); returned as a 4th list
    """The graph GraphPPL builds for the benchmark notebook's model (cell 4) / mlgssm_test.jl:9-17.  X_of_t(t): the constant
    of time index t when the @model loop indexes an array of matrices (`A[t] * x[t-1]`, `Σ = P[t]`, …)."""
    A, B = np.asarray(A, float), np.asarray(B, float)
    d, dy = A.shape[0], B.shape[0]
    gb = GraphBuilder()
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(m0), gb.constvar(V0))
    xs, ys, us = [], [], []
    for t in range(T):
        if t > 0 or prior_through_transition:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A if A_of_t is None else A_of_t(t)), x)
            if c_of_t is not None and c_of_t(t) is not None:   # `A * x[t-1] + c[t]`: a known input
                w = gb.randomvar(d)
                gb.node(_lib.NODE_ADD, w, *((gb.constvar(c_of_t(t)), a) if const_first else (a, gb.constvar(c_of_t(t)))))
                a = w
            if du:                                             # `… + B_u * u[t]` with u[t] a data variable
                u = gb.datavar(du)
                us.append(u)
                if Bu is not None:
                    bu = gb.randomvar(d)
                    gb.multiply(bu, gb.constvar(Bu), u)
                    u = bu
                w = gb.randomvar(d)
                gb.node(_lib.NODE_ADD, w, a, u)
                a = w
            xn = gb.randomvar(d)
            gb.mvnormal_mean_cov(xn, a, gb.constvar(P if P_of_t is None else P_of_t(t)))
            x = xn
        b = gb.randomvar(dy)
        gb.multiply(b, gb.constvar(B if B_of_t is None else B_of_t(t)), x)
        if d_of_t is not None and d_of_t(t) is not None:       # `B * x[t] + d[t]`
            w = gb.randomvar(dy)
            gb.node(_lib.NODE_ADD, w, b, gb.constvar(d_of_t(t)))
            b = w
        y = gb.datavar(dy)
        gb.mvnormal_mean_cov(y, b, gb.constvar(Q if Q_of_t is None else Q_of_t(t)))
        xs.append(x); ys.append(y)
    if du:
        return gb, xs, ys, us
    return gb, xs, ys


def two_branch_chain_graph(T, A, B1, B2, P, Q1, Q2, m0, V0):
    """x[t] ~ MvNormal(A x[t-1], P) with TWO observation branches per state, y1[t] ~ MvNormal(B1 x[t], Q1), y2[t] ~ MvNormal(B2 x[t], Q2) —
    a graph the state-space pattern matcher rejects (csrc/graph_lowering.hpp) and the node-array executor runs.  Returns (builder, state
    variables, data variables in time order y1[0], y2[0], y1[1], …)."""
    gb = GraphBuilder()
    d = np.asarray(A).shape[0]
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(m0), gb.constvar(V0))
    xs, ys = [], []
    for t in range(T):
        if t:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A), x)
            xn = gb.randomvar(d)
            gb.mvnormal_mean_cov(xn, a, gb.constvar(P))
            x = xn
        for B, Q in ((B1, Q1), (B2, Q2)):
            B = np.asarray(B, float)
            b = gb.randomvar(B.shape[0])
            gb.multiply(b, gb.constvar(B), x)
            y = gb.datavar(B.shape[0])
            gb.mvnormal_mean_cov(y, b, gb.constvar(Q))
            ys.append(y)
        xs.append(x)
    return gb, xs, ys


def lgssm_noise_graph(T, A, B, P, m0, V0, nu0, S0, init=None, prior_through_transition=False, gamma=None):
    """The chain of `lgssm_graph` with an unknown observation-noise precision: `W ~ Wishart(nu0, S0)` and every observation node
    `y[t] ~ MvNormal(μ = B * x[t], Λ = W)` (test/models/iid/mv_iid_precision_tests.jl:11-15 spells the node pair).  init = (nu, V): the
    `@initialization` marginal q(W).  gamma = "rate" | "scale" (dy = 1): `τ ~ Gamma(shape = nu0, rate | scale = S0)` with
    `y[t] ~ Normal(mean = …, precision = τ)` and init = (shape, rate).  Returns (builder, state vars, data vars, W)."""
    A, B = np.asarray(A, float), np.asarray(B, float)
    d, dy = A.shape[0], B.shape[0]
    gb = GraphBuilder()
    W = gb.randomvar(dy, name="W")
    if gamma:
        gb.node(_lib.NODE_GAMMA_SHAPE_RATE if gamma == "rate" else _lib.NODE_GAMMA_SHAPE_SCALE, W, gb.constvar(float(nu0)), gb.constvar(float(S0)))
    else:
        gb.node(_lib.NODE_WISHART, W, gb.constvar(float(nu0)), gb.constvar(np.asarray(S0, float).reshape(dy, dy)))
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(m0), gb.constvar(V0))
    xs, ys = [], []
    for t in range(T):
        if t > 0 or prior_through_transition:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A), x)
            xn = gb.randomvar(d)
            gb.mvnormal_mean_cov(xn, a, gb.constvar(P))
            x = xn
        b = gb.randomvar(dy)
        gb.multiply(b, gb.constvar(B), x)
        y = gb.datavar(dy)
        gb.node(_lib.NODE_NORMAL_MEAN_PRECISION if gamma else _lib.NODE_MVNORMAL_MEAN_PRECISION, y, b, W)
        xs.append(x); ys.append(y)
    if init is not None:
        if gamma:
            gb.initialize(W, _lib.INIT_GAMMA, np.asarray(init, float))
        else:
            gb.initialize(W, _lib.INIT_WISHART, np.concatenate([[float(init[0])], np.ravel(np.asarray(init[1], float))]))
    return gb, xs, ys, W


def lower_lgssm_noise(g):
    """Host-only lowering of the chain with an unknown observation-noise precision: dict(d, dy, T, A, B, P, m0, V0, nu0, S0, init_nu, init_V, …)."""
    L = _lib.lib()
    out = _lib.LgssmNoiseLowered()
    st = L.rxhip_graph_lower_lgssm_noise(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    d, dy, T = out.chain.d, out.chain.dy, out.chain.T
    bufs = dict(A=np.empty((d, d)), B=np.empty((dy, d)), P=np.empty((d, d)), m0=np.empty(d), V0=np.empty((d, d)))
    for k, v in bufs.items():
        setattr(out.chain, k, v.ctypes.data_as(_lib.c_double_p))
    sv, dv = np.empty(T, dtype=np.int64), np.empty(T, dtype=np.int64)
    out.chain.state_var = sv.ctypes.data_as(_lib.c_int64_p)
    out.chain.data_var = dv.ctypes.data_as(_lib.c_int64_p)
    S0, iV = np.empty((dy, dy)), np.empty((dy, dy))
    out.S0, out.init_V = S0.ctypes.data_as(_lib.c_double_p), iV.ctypes.data_as(_lib.c_double_p)
    st = L.rxhip_graph_lower_lgssm_noise(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    return dict(d=d, dy=dy, T=T, prior_through_transition=bool(out.chain.prior_through_transition), state_var=sv, data_var=dv,
                precision_var=int(out.precision_var), nu0=float(out.nu0), S0=S0, init_nu=float(out.init_nu), init_V=iV, **bufs)


def create_noise_engine_from_graph(g, segments=0, device=-1, stream=None):
    """rxhip_create on a chain with an unknown observation-noise precision: an LGSSMNoiseEngine around the new handle."""
    from .engine import LGSSMNoiseEngine

    low = lower_lgssm_noise(g)
    L = _lib.lib()
    h = ctypes.c_void_p()
    st = L.rxhip_create(ctypes.byref(g), int(segments), int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
    if st != _lib.OK:
        msg = L.rxhip_lowering_error().decode() or (L.rxhip_last_error(h).decode() if h else "")
        if h:
            L.rxhip_destroy(h)
        raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
    eng = LGSSMNoiseEngine.__new__(LGSSMNoiseEngine)
    eng._h, eng.d, eng.dy, eng.T, eng.n_chains, eng.n_models = h, low["d"], low["dy"], low["T"], int(g.n_replicas or 1), 1
    eng.horizon, eng.du = 0, 0
    eng._keep, eng._data_ref, eng._iters = [], None, 0
    return eng


def scalar_chain_graph(T, a, b, p, q, m0, v0, prior_through_transition=False, spell="normal", precision=False):
    """Scalar random-walk / AR(1) chains in the spellings RxInfer users write them:
    spell = "normal":  x[t] ~ Normal(mean = x[t-1], var = p), y[t] ~ Normal(mean = x[t], var = q)   (a = b = 1, no `*` nodes)
    spell = "scaled":  x[t] ~ Normal(mean = a * x[t-1], var = p), y[t] ~ Normal(mean = b * x[t], var = q)
    spell = "mixed":   `*` on the transition only.
    precision = True: every Gaussian node is `NormalMeanPrecision(μ, 1/var)` (test/inference/prediction_tests.jl:197-213)."""
    gb = GraphBuilder()
    x = gb.randomvar(1)
    if precision:
        normal = lambda out, mu, var: gb.node(_lib.NODE_NORMAL_MEAN_PRECISION, out, mu, gb.constvar(1.0 / var))
    else:
        normal = lambda out, mu, var: gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, out, mu, gb.constvar(var))
    normal(x, gb.constvar(m0), v0)
    xs, ys = [], []
    for t in range(T):
        if t > 0 or prior_through_transition:
            mu = x
            if spell in ("scaled", "mixed"):
                mu = gb.randomvar(1)
                gb.multiply(mu, gb.constvar(a), x)
            xn = gb.randomvar(1)
            normal(xn, mu, p)
            x = xn
        mu = x
        if spell == "scaled":
            mu = gb.randomvar(1)
            gb.multiply(mu, gb.constvar(b), x)
        y = gb.datavar(1)
        normal(y, mu, q)
        xs.append(x); ys.append(y)
    return gb, xs, ys


def drift_chain_graph(T, m0, v0, c, obs_var, const_first=False):
    """test/models/statespace/ulgssm_tests.jl:8-15: x_prior ~ Normal(μ, v); x[i] ~ x_prev + c; y[i] ~ Normal(μ = x[i], v = P)."""
    gb = GraphBuilder()
    x = gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, gb.constvar(m0), gb.constvar(v0))
    xs, ys = [], []
    for _ in range(T):
        xn = gb.randomvar(1)
        if const_first:
            gb.node(_lib.NODE_ADD, xn, gb.constvar(c), x)
        else:
            gb.node(_lib.NODE_ADD, xn, x, gb.constvar(c))
        x = xn
        y = gb.datavar(1)
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, x, gb.constvar(obs_var))
        xs.append(x); ys.append(y)
    return gb, xs, ys


def lowering_asymmetry():
    """max|W − W′| / max|W| of the least symmetric constant parameter the last lowering call of this thread accepted (and symmetrised); 0.0: none."""
    return float(_lib.lib().rxhip_lowering_asymmetry())


def lower_lgssm(g):
    """Host-only lowering (no GPU): returns dict(d, dy, T, prior_through_transition, A, B, P, Q, m0, V0, state_var, data_var)."""
    L = _lib.lib()
    out = _lib.LgssmLowered()
    st = L.rxhip_graph_lower_lgssm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    d, dy, T, M = out.d, out.dy, out.T, out.n_models
    bufs = dict(A=np.empty((M, d, d)), B=np.empty((M, dy, d)), P=np.empty((M, d, d)), Q=np.empty((M, dy, dy)), m0=np.empty(d),
                V0=np.empty((d, d)), c=np.empty(d))
    sv, dv, sm = np.empty(T, dtype=np.int64), np.empty(T, dtype=np.int64), np.empty(T, dtype=np.int32)
    bufs["state_offset"], bufs["obs_offset"] = np.empty((T, d)), np.empty((T, dy))
    for k, v in bufs.items():
        setattr(out, k, v.ctypes.data_as(_lib.c_double_p))
    out.state_var = sv.ctypes.data_as(_lib.c_int64_p)
    out.data_var = dv.ctypes.data_as(_lib.c_int64_p)
    out.step_model = sm.ctypes.data_as(_lib.c_int32_p)
    du = int(out.du)
    Bu, uv = np.empty((d, max(du, 1))), np.empty(T, dtype=np.int64)
    out.input_matrix = Bu.ctypes.data_as(_lib.c_double_p)
    out.input_var = uv.ctypes.data_as(_lib.c_int64_p)
    st = L.rxhip_graph_lower_lgssm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    bufs["input_matrix"], bufs["input_var"], bufs["du"] = (Bu if du else None), (uv if du else None), du
    if M == 1:  # time-invariant: plain matrices, as before
        for k in "ABPQ":
            bufs[k] = bufs[k][0]
    return dict(d=d, dy=dy, T=T, prior_through_transition=bool(out.prior_through_transition), deterministic=bool(out.deterministic),
                state_var=sv, data_var=dv, n_models=M, step_model=sm if M > 1 else None, has_offsets=bool(out.has_offsets), **bufs)


def create_engine_from_graph(g, segments=0, device=-1, stream=None):
    """rxhip_create: lower + build the engine; returns a raw handle wrapped as an LGSSMEngine-compatible object."""
    from .engine import LGSSMEngine

    L = _lib.lib()
    h = ctypes.c_void_p()
    st = L.rxhip_create(ctypes.byref(g), int(segments), int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
    if st != _lib.OK:
        msg = L.rxhip_lowering_error().decode() or (L.rxhip_last_error(h).decode() if h else "")
        if h:
            L.rxhip_destroy(h)
        raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
    info = _lib.TreeInfo()
    if L.rxhip_tree_get_info(h, ctypes.byref(info)) == _lib.OK:   # rxhip_create handed the graph to the node-array executor (e.g. a chain beyond the conditioning
        why = L.rxhip_lowering_error().decode()                   # envelope of the information-form engines): that handle answers to rxhip_tree_*, not to this wrapper
        L.rxhip_destroy(h)
        raise RxHipError(_lib.ERR_UNSUPPORTED, "no pattern-matched engine took this graph (rxhip.tree.TreeEngine(gb, force_executor=False) runs it on the executor)" + (": " + why if why else ""))
    low = lower_lgssm(g)
    if low["deterministic"]:
        from .engine import DriftChainEngine

        eng = DriftChainEngine.__new__(DriftChainEngine)
    else:
        eng = LGSSMEngine.__new__(LGSSMEngine)
    eng._h, eng.d, eng.dy, eng.T, eng.n_chains, eng.n_models = h, low["d"], low["dy"], low["T"], int(g.n_replicas or 1), low["n_models"]
    eng.horizon = 0
    eng.du = int(low["du"])
    eng._keep, eng._data_ref, eng._iters = [], None, 0
    return eng


def mixture_graph(N, prior_mean, prior_var, prior_shape, prior_rate, prior_alpha, init=None, bernoulli=False):
    """The graph of `univariate_gaussian_mixture_model` (test/models/mixtures/gmm_univariate_tests.jl:7-20) with K
    components (`Beta` / `Bernoulli` for the reference's K = 2 spelling, else `Dirichlet` / `Categorical`).
    init = dict(m=(means, vars), p=(shapes, rates), s=alphas) are the `@initialization` marginals."""
    K = len(prior_mean)
    gb = GraphBuilder()
    s = gb.randomvar(K)
    if bernoulli:
        assert K == 2
        gb.node(_lib.NODE_BETA, s, gb.constvar(prior_alpha[0]), gb.constvar(prior_alpha[1]))
    else:
        gb.node(_lib.NODE_DIRICHLET, s, gb.constvar(np.asarray(prior_alpha, float)))
    m, p = [], []
    for k in range(K):
        mk, pk = gb.randomvar(1), gb.randomvar(1)
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, mk, gb.constvar(prior_mean[k]), gb.constvar(prior_var[k]))
        gb.node(_lib.NODE_GAMMA_SHAPE_RATE, pk, gb.constvar(prior_shape[k]), gb.constvar(prior_rate[k]))
        m.append(mk); p.append(pk)
    ys = []
    for _ in range(N):
        z, y = gb.randomvar(1), gb.datavar(1)
        gb.node(_lib.NODE_BERNOULLI if bernoulli else _lib.NODE_CATEGORICAL, z, s)
        gb.node(_lib.NODE_NORMAL_MIXTURE, y, z, *m, *p)
        ys.append(y)
    if init is not None:
        for k in range(K):
            gb.initialize(m[k], _lib.INIT_NORMAL, (init["m"][0][k], init["m"][1][k]))
            gb.initialize(p[k], _lib.INIT_GAMMA, (init["p"][0][k], init["p"][1][k]))
        if "s" in init:
            gb.initialize(s, _lib.INIT_DIRICHLET, init["s"])
    return gb, ys


def mv_mixture_graph(N, prior_mean, prior_cov, prior_nu, prior_scale, prior_alpha, init=None):
    """The graph of `multivariate_gaussian_mixture_model` (test/models/mixtures/gmm_multivariate_tests.jl:6-32).
    init = dict(m=(means [K][d], covs [K][d][d]), w=(nus [K], scales [K][d][d]), s=alphas)."""
    prior_mean = np.asarray(prior_mean, float)
    K, d = prior_mean.shape
    gb = GraphBuilder()
    s = gb.randomvar(K)
    gb.node(_lib.NODE_DIRICHLET, s, gb.constvar(np.asarray(prior_alpha, float)))
    m, w = [], []
    for k in range(K):
        mk, wk = gb.randomvar(d), gb.randomvar(d)
        gb.node(_lib.NODE_MVNORMAL_MEAN_COV, mk, gb.constvar(prior_mean[k]), gb.constvar(np.asarray(prior_cov[k], float)))
        gb.node(_lib.NODE_WISHART, wk, gb.constvar(float(prior_nu[k])), gb.constvar(np.asarray(prior_scale[k], float)))
        m.append(mk); w.append(wk)
    ys = []
    for _ in range(N):
        z, y = gb.randomvar(1), gb.datavar(d)
        gb.node(_lib.NODE_CATEGORICAL, z, s)
        gb.node(_lib.NODE_NORMAL_MIXTURE, y, z, *m, *w)
        ys.append(y)
    if init is not None:
        for k in range(K):
            gb.initialize(m[k], _lib.INIT_MVNORMAL, np.concatenate([np.ravel(init["m"][0][k]), np.ravel(init["m"][1][k])]))
            gb.initialize(w[k], _lib.INIT_WISHART, np.concatenate([[init["w"][0][k]], np.ravel(init["w"][1][k])]))
        if "s" in init:
            gb.initialize(s, _lib.INIT_DIRICHLET, init["s"])
    return gb, ys


def mv_iid_graph(N, prior_mean, prior_precision, prior_nu, prior_scale, init=None):
    """The graph of `mv_iid_wishart` (test/models/iid/mv_iid_precision_tests.jl:11-15): m ~ MvNormal(μ, Λ); P ~ Wishart(ν, S);
    y[i] ~ MvNormal(μ = m, Λ = P).  init = dict(m=(mean [d], cov [d][d]), w=(nu, scale [d][d])): the `@initialization` marginals."""
    prior_mean = np.asarray(prior_mean, float)
    d = prior_mean.shape[0]
    gb = GraphBuilder()
    m, P = gb.randomvar(d), gb.randomvar(d)
    gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, m, gb.constvar(prior_mean), gb.constvar(np.asarray(prior_precision, float)))
    gb.node(_lib.NODE_WISHART, P, gb.constvar(float(prior_nu)), gb.constvar(np.asarray(prior_scale, float)))
    ys = []
    for _ in range(N):
        y = gb.datavar(d)
        gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, y, m, P)
        ys.append(y)
    if init is not None:
        gb.initialize(m, _lib.INIT_MVNORMAL, np.concatenate([np.ravel(init["m"][0]), np.ravel(init["m"][1])]))
        gb.initialize(P, _lib.INIT_WISHART, np.concatenate([[init["w"][0]], np.ravel(init["w"][1])]))
    return gb, ys


def lower_mvgmm(g):
    """Host-only lowering of a multivariate mixture graph."""
    L = _lib.lib()
    out = _lib.MvGmmLowered()
    st = L.rxhip_graph_lower_mvgmm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    N, K, d = out.N, out.K, out.d
    shapes = dict(mu0=(K, d), S0=(K, d, d), nu0=(K,), V0=(K, d, d), alpha0=(K,), init_m_mean=(K, d), init_m_cov=(K, d, d),
                  init_w_nu=(K,), init_w_V=(K, d, d), init_s_alpha=(K,))
    bufs = {n: np.empty(sh) for n, sh in shapes.items()}
    dv = np.empty(N, dtype=np.int64)
    for n, v in bufs.items():
        setattr(out, n, v.ctypes.data_as(_lib.c_double_p))
    out.data_var = dv.ctypes.data_as(_lib.c_int64_p)
    st = L.rxhip_graph_lower_mvgmm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    return dict(N=N, K=K, d=d, data_var=dv, **bufs)


def iid_normal_graph(N, mean, variance, shape, rate=None, init=None, scale=None):
    """`iid_gaussians_params` (test/models/models_tests.jl:114-128): m ~ Normal, p ~ Gamma, y[i] ~ Normal(mean = m, precision = p).
    `scale = θ` emits the reference's own node for that model, `Gamma(shape = …, scale = …)` -> GammaShapeScale."""
    gb = GraphBuilder()
    m, p = gb.randomvar(1), gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, m, gb.constvar(mean), gb.constvar(variance))
    if scale is None:
        gb.node(_lib.NODE_GAMMA_SHAPE_RATE, p, gb.constvar(shape), gb.constvar(rate))
    else:
        gb.node(_lib.NODE_GAMMA_SHAPE_SCALE, p, gb.constvar(shape), gb.constvar(scale))
    ys = []
    for _ in range(N):
        y = gb.datavar(1)
        gb.node(_lib.NODE_NORMAL_MEAN_PRECISION, y, m, p)
        ys.append(y)
    if init is not None:
        gb.initialize(m, _lib.INIT_NORMAL, init["m"])
        gb.initialize(p, _lib.INIT_GAMMA, init["p"])
    return gb, ys


def hgf_step_graph(kappa, omega, z_variance, y_variance, q_zt=(0.0, 5.0), q_xt=(0.0, 5.0), n_gh=31):
    """The one-step graph of test/models/statespace/hgf_tests.jl:9-31 with its `@initialization` and GCV meta."""
    gb = GraphBuilder()
    zt_min, xt_min, zt, xt = gb.randomvar(1), gb.randomvar(1), gb.randomvar(1), gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, zt_min, gb.datavar(1), gb.datavar(1))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, xt_min, gb.datavar(1), gb.datavar(1))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, zt, zt_min, gb.constvar(z_variance))
    gb.node(_lib.NODE_GCV, xt, xt_min, zt, gb.constvar(kappa), gb.constvar(omega))
    y = gb.datavar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, xt, gb.constvar(y_variance))
    gb.initialize(zt, _lib.INIT_NORMAL, q_zt)
    gb.initialize(xt, _lib.INIT_NORMAL, q_xt)
    gb.gh_points = n_gh
    return gb, dict(zt=zt, xt=xt, y=y)


def lower_gmm(g):
    """Host-only lowering of a mixture graph: dict(N, K, mu0, v0, a0, b0, alpha0, init_*, data_var)."""
    L = _lib.lib()
    out = _lib.GmmLowered()
    st = L.rxhip_graph_lower_gmm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    N, K = out.N, out.K
    names = ("mu0", "v0", "a0", "b0", "alpha0", "init_m_mean", "init_m_var", "init_p_shape", "init_p_rate", "init_s_alpha")
    bufs = {n: np.empty(K) for n in names}
    dv = np.empty(N, dtype=np.int64)
    for n, v in bufs.items():
        setattr(out, n, v.ctypes.data_as(_lib.c_double_p))
    out.data_var = dv.ctypes.data_as(_lib.c_int64_p)
    st = L.rxhip_graph_lower_gmm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    return dict(N=N, K=K, data_var=dv, **bufs)


def lower_hgf(g):
    L = _lib.lib()
    out = _lib.HgfLowered()
    st = L.rxhip_graph_lower_hgf(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    return {n: getattr(out, n) for n, _ in _lib.HgfLowered._fields_}


def create_vmp_engine_from_graph(g, device=-1, stream=None):
    """rxhip_create for the mean-field families: returns a GMMEngine / HGFEngine around the new handle."""
    from .engine import GMMEngine, HGFEngine

    L = _lib.lib()
    h = ctypes.c_void_p()
    st = L.rxhip_create(ctypes.byref(g), 0, int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
    if st != _lib.OK:
        msg = L.rxhip_lowering_error().decode() or (L.rxhip_last_error(h).decode() if h else "")
        if h:
            L.rxhip_destroy(h)
        raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
    if any(t == _lib.NODE_GCV for t in np.ctypeslib.as_array(g.factor_type, (g.n_factors,))):
        eng = HGFEngine.__new__(HGFEngine)
        eng._h, eng.T, eng.n_series, eng._iters = h, int(g.n_observations), int(g.n_replicas or 1), 0
        eng.n_chains = eng.n_series
    elif any(t == _lib.NODE_WISHART for t in np.ctypeslib.as_array(g.factor_type, (g.n_factors,))):
        from .engine import MvGMMEngine

        low = lower_mvgmm(g)
        eng = MvGMMEngine.__new__(MvGMMEngine)
        eng._h, eng.N, eng.K, eng.d, eng._iters = h, low["N"], low["K"], low["d"], 0
    else:
        low = lower_gmm(g)
        eng = GMMEngine.__new__(GMMEngine)
        eng._h, eng.N, eng.K, eng._iters = h, low["N"], low["K"], 0
    eng._keep, eng._data_ref = [], None
    return eng
