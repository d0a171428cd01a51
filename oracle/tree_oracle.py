"""Generic sum-product / mean-field VMP on an arbitrary acyclic Gaussian factor graph — CPU restatement, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product (rxinfer.jl_amd/) never does.

What it restates.  The reference materialises ANY GraphPPL model as `factornode(fform, interfaces, factorization)` objects
(/root/reference/src/model/plugins/reactivemp_inference.jl:490-540), multiplies the inbound messages of every random variable from left to
right (:365-374, :432-447), forms marginals as the product of all inbound messages (:440-447) and sums the Bethe free energy over node and
variable terms (src/model/plugins/reactivemp_free_energy.jl:51-126).  The rule bodies live in ReactiveMP.jl ~6.0 / ExponentialFamily.jl 2.1
(un-vendored; SURVEY.md Appendix A restates them); the ones used here:

  MvNormalMeanCovariance / NormalMeanVariance (out, μ, Σ)   :out  N(mean(m_μ), cov(m_μ) + Σ)         :μ  N(mean(m_out), cov(m_out) + Σ)
  MvNormalMeanPrecision / NormalMeanPrecision (out, μ, Λ)   the same with Σ = Λ⁻¹ (constant Λ) or Σ = mean(q_Λ)⁻¹ (q(out, μ) q(Λ), mean field)
        toward Λ:  Wishart(d + 2, E[(out − μ)(out − μ)ᵀ]⁻¹)   (scalar: Gamma(3/2, ½ E[(out − μ)²]))
  typeof(*) (out, A, in), A constant      :out  N(A m, A V Aᵀ)                    :in  (Aᵀ ξ, Aᵀ Λ A) in weighted-mean / precision form
  typeof(+) (out, in1, in2)               :out  N(m1 + m2, V1 + V2)               :in1 N(m_out − m2, V_out + V2)   (a constant / data input: V = 0)
  Wishart (out, ν, S), GammaShapeRate / GammaShapeScale (out, α, β | θ): constant parameters, the prior of a precision variable
  product of Gaussians: (ξ1 + ξ2, Λ1 + Λ2);  marginal = product of all inbound messages
  NormalMixture (out, switch, m[1..K], p[1..K]) under MeanField() with  switch ~ Categorical(s),  s ~ Dirichlet(a) | constant  (round 6; the rules of
        oracle/rxoracle.c rxo_mvgmm_vmp, node by node instead of summed over the data set): the node's log-factor is Σ_k z_k log N(out | m_k, p_k⁻¹), so under
        mean field it acts on (out, m_k, p_k) as K Gaussian precision nodes each WEIGHTED by π_k = q(z = k) — toward m_k: (π_k E[p_k] E[out], π_k E[p_k]),
        toward out: the same with E[m_k], toward p_k: ν += π_k, V⁻¹ += π_k E[(out − m_k)(out − m_k)ᵀ], energy Σ_k π_k U_k — and toward the switch:
        q(z = k) ∝ exp(E log s_k − U_k), U_k = ½[d log 2π − E log|p_k| + tr(E[p_k] E[(out − m_k)(out − m_k)ᵀ])];  q(s) = Dirichlet(a + Σ_i π_i).
        Schedule per iteration (rxo_mvgmm_vmp): q(z) from the previous marginals, then the Gaussian sweep and q(s) with the new π, then q(p) with the new q(m).
  GCV (y, x, z, κ, ω), κ and ω constants, under q(y, x) q(z) (test/models/statespace/hgf_tests.jl:28-35; the rules of oracle/rxoracle.c rxo_hgf_filter): toward
        (y, x) the node IS a scalar Gaussian precision node with γ = exp(−(κ z + ω)): E[γ] = exp(−ω − κ m_z + κ² v_z / 2), E[log γ] = −(κ m_z + ω) from q(z) of the
        previous iteration.  Toward z it sends ExponentialLinearQuadratic(a = κ, b = ψ e^{−ω}, c = −κ), ψ = E[(y − x)²] under the node-local joint; q(z) is its product
        with the Gaussian message from the rest of the graph, moment-matched by Gauss–Hermite cubature against that message; every OTHER neighbour of z sees the
        message through its Gaussian moments (cubature of pdf(z)·exp(z²/2) against N(0, 1)) — the reference's treatment, reproduced to the golden by rxo_hgf_filter.
  Bethe terms: SURVEY Appendix A.4 as oracle/rxoracle.c lgssm_bp_ws spells them for the state-space graph — stochastic node U − H[q(cluster)],
  deterministic node −H[q(inputs)], random variable (degree − 1) H[q]; clamped interfaces contribute no entropy.

One deviation, stated: the additive rule applied to a message in weighted-mean / precision form uses Λ' = Λ (Λ + W)⁻¹ W, ξ' = W (Λ + W)⁻¹ ξ
(W = Σ⁻¹) instead of inverting Λ first — the same message wherever the reference's `mean_cov` exists, and defined for the rank-deficient
backward messages of an observation map with fewer rows than columns (the reference's cholinv throws there).  The backward rule of `+` with two
random inputs is the same rule with the other input as the noise: a precision-form message from `out` gives Λ' = Λo (Λo + W2)⁻¹ W2,
ξ' = W2 (Λo + W2)⁻¹ (ξo + ξ2) − ξ2 (= N(m_out − m2, V_out + V2) wherever V_out exists): `y ~ N(b'(x + u), q)` with a row vector b is inside the family.

Pinning of the generic machinery on RANDOM graphs: tests/tree_graphs.py::random_forest grows acyclic graphs out of every construct above;
tests/test_tree_oracle.py::test_random_forests_against_brute_force holds this module to the conditioned joint Gaussian on 24 of them.

Pinning.  tests/test_tree_oracle.py checks this module against (i) brute-force conditioning of the joint Gaussian (marginals, and
free energy = −log evidence) on random trees, (ii) oracle/rxoracle.c's lgssm_bp / lgssm_noise_vmp on the state-space graphs, which are pinned to
the reference's golden free energies (tests/test_golden_reference.py), (iii) the RNG-free known answers 3.51551 / 2.26551
(/root/reference/test/models/models_tests.jl:255,308).

Input: a graph in the exchange format "rxhip-graph-1" (what HIPInferencePlugin.jl's dump_graph writes; a dict) and the data of ONE replica,
{variable id: vector}."""
import math

import numpy as np
from scipy.special import digamma, gammaln

LOG2PI = math.log(2.0 * math.pi)

GAUSS_COV = ("MvNormalMeanCovariance", "NormalMeanVariance")
GAUSS_PREC = ("MvNormalMeanPrecision", "NormalMeanPrecision")
PRIORS = ("Wishart", "GammaShapeRate", "GammaShapeScale")
SKIP = ("NormalMixture", "Categorical", "Dirichlet", "Bernoulli", "Beta", "GCV")   # no Gaussian message of their own: the mixture node acts through its K weighted virtual nodes


def _sym(M):
    return 0.5 * (M + M.T)


class ImproperMessage(ValueError):
    """A rule asked for the other form of a message whose matrix is singular (a rank-deficient precision wanted as a covariance: the reference's `mean_cov` of a
    MvNormalWeightedMeanPrecision throws a PosDefException there).  numpy.linalg.inv returns numbers of size 1e16 for such a matrix more often than it raises;
    the check is the residual of the computed inverse."""


def _inverse(B):
    try:
        X = np.linalg.inv(B)
    except np.linalg.LinAlgError as e:
        raise ImproperMessage(str(e))
    if not np.all(np.isfinite(X)) or np.max(np.abs(B @ X - np.eye(B.shape[0]))) > 1e-3:
        raise ImproperMessage("a message's matrix is singular to working precision")
    return X


class Msg:
    """A Gaussian message in moment form (m, V) or weighted-mean / precision form (xi, L)."""

    def __init__(self, form, a, B):
        self.form, self.a, self.B = form, np.asarray(a, float), np.asarray(B, float)

    def mv(self):
        if self.form == "mv":
            return self.a, self.B
        V = _inverse(self.B)
        return V @ self.a, _sym(V)

    def wp(self):
        if self.form == "wp":
            return self.a, self.B
        L = _inverse(self.B)
        return L @ self.a, _sym(L)


def _entropy(V):
    d = V.shape[0]
    return 0.5 * (d * (LOG2PI + 1.0) + np.linalg.slogdet(V)[1])


def mvdigamma(a, d):
    return sum(digamma(a - 0.5 * i) for i in range(d))


def mvlgamma(a, d):
    return 0.25 * d * (d - 1) * math.log(math.pi) + sum(gammaln(a - 0.5 * i) for i in range(d))


class TreeGraph:
    def __init__(self, dump):
        if dump.get("format") != "rxhip-graph-1":
            raise ValueError("not an rxhip-graph-1 dump")
        self.vars = dump["variables"]
        self.factors = [(f["type"], [int(v) for _, v in f["interfaces"]]) for f in dump["factors"]]
        self.clusters = [f.get("clusters") for f in dump["factors"]]   # the node's factorisation of q (VariationalConstraintsFactorizationIndicesKey), None: the default
        nv = len(self.vars)
        # NormalMixture under mean field = K weighted Gaussian precision nodes (out, m_k, p_k) appended behind the real factors; the mixture node itself, the
        # Categorical and the Dirichlet nodes carry no Gaussian message (types in SKIP)
        self.weight, self.mixtures, self.cat, self.dir, self.gcv = {}, [], {}, {}, []
        self.gh_points = int(dump.get("gh_points") or 31)
        for fi, (t, ifs) in enumerate(list(self.factors)):
            if t == "NormalMixture":
                K = (len(ifs) - 2) // 2
                if len(ifs) != 2 + 2 * K or K < 1:
                    raise ValueError("NormalMixture: (out, switch, m[1..K], p[1..K]) expected")
                cl = self.clusters[fi]
                rnd = [k for k, v in enumerate(ifs) if self.vars[v]["kind"] == "random"]
                if cl is not None and len({cl[k] for k in rnd}) != len(rnd):
                    raise ValueError("NormalMixture: only the mean-field factorisation has rules")
                vf = []
                for k in range(K):
                    self.weight[len(self.factors)] = (ifs[1], k)
                    vf.append(len(self.factors))
                    self.factors.append(("MvNormalMeanPrecision", [ifs[0], ifs[2 + k], ifs[2 + K + k]]))
                    self.clusters.append([0, 1, 2])
                self.mixtures.append(dict(node=fi, out=ifs[0], z=ifs[1], m=ifs[2:2 + K], p=ifs[2 + K:], virtual=vf))
            elif t == "GCV":
                if len(ifs) != 5 or any(int(self.vars[v]["rows"]) != 1 for v in ifs[:3]) or any(self.vars[v]["kind"] != "constant" for v in ifs[3:]):
                    raise ValueError("GCV: scalar (y, x, z) and constant (κ, ω) expected")
                cl = self.clusters[fi]
                if cl is not None and (cl[0] != cl[1] or cl[2] == cl[0]):
                    raise ValueError("GCV: only q(y, x) q(z) has rules")
                gam = len(self.vars)   # the node's precision γ(z): a virtual variable, neither random nor clamped
                self.vars = list(self.vars) + [dict(name=f"gcv_gamma_{fi}", kind="gcvprec", rows=1, cols=1)]
                fa, fz = len(self.factors), len(self.factors) + 1
                self.factors.append(("NormalMeanPrecision", [ifs[0], ifs[1], gam]))
                self.clusters.append([0, 0, 1])
                self.factors.append(("GCVZ", [ifs[2]]))
                self.clusters.append([0])
                self.gcv.append(dict(node=fi, y=ifs[0], x=ifs[1], z=ifs[2], kappa=float(np.ravel(self.vars[ifs[3]]["value"])[0]), omega=float(np.ravel(self.vars[ifs[4]]["value"])[0]),
                                     fa=fa, fz=fz, gamma=gam))
            elif t in ("Categorical", "Bernoulli"):   # Bernoulli(s): the two-component spelling, z = true the FIRST component
                self.cat[ifs[0]] = ifs[1]
            elif t == "Dirichlet":
                self.dir[ifs[0]] = ifs[1]
            elif t == "Beta":                          # Beta(a, b) on the probability of the first component = Dirichlet([a, b])
                self.dir[ifs[0]] = (ifs[1], ifs[2])
        nv = len(self.vars)
        self.dim = [int(v["rows"]) for v in self.vars]
        self.kind = [v["kind"] for v in self.vars]
        # classify: precision variables are the random `out` of a Wishart / Gamma prior node
        self.prec_prior = {}
        for fi, (t, ifs) in enumerate(self.factors):
            if t in PRIORS:
                self.prec_prior[ifs[0]] = fi
        # the output of a deterministic node whose inputs are all clamped (constants, data, such outputs) is clamped itself: `a + b` of two
        # data variables is a PointMass message in the reference (test/models/models_tests.jl:242-256), not a random variable
        self.derived = {}
        changed = True
        while changed:
            changed = False
            for fi, (t, ifs) in enumerate(self.factors):
                if t in ("*", "+") and ifs[0] not in self.derived and self.kind[ifs[0]] == "random" and \
                        all(self.kind[x] != "random" or x in self.derived for x in ifs[1:]):
                    self.derived[ifs[0]] = fi
                    changed = True
        self.gauss = [self.kind[v] == "random" and v not in self.prec_prior and v not in self.derived and v not in self.cat and v not in self.dir for v in range(nv)]
        self.nbrs = [[] for _ in range(nv)]   # per variable: (factor, interface) in factor order — the fold order of the product
        for fi, (t, ifs) in enumerate(self.factors):
            if t in SKIP:
                continue
            for k, v in enumerate(ifs):
                self.nbrs[v].append((fi, k))
        # Gaussian nodes the constraints run under q(out) q(μ): mean field between the two Gaussian interfaces
        self.mf = [t in GAUSS_COV + GAUSS_PREC and cl is not None and self.gauss[ifs[0]] and self.gauss[ifs[1]] and cl[0] != cl[1]
                   for (t, ifs), cl in zip(self.factors, self.clusters)]
        self._check_supported()

    def init_gauss(self, v):
        """the `@initialization` marginal of a Gaussian variable as (mean, covariance)"""
        ini = self.vars[v].get("init")
        d = self.dim[v]
        if ini is not None and ini["family"] in ("normal", "mvnormal"):
            p = np.asarray(ini["params"], float)
            return p[:d].copy(), p[d:d + d * d].reshape(d, d).copy()
        # the anonymous output of `A * x` cannot be named in an @initialization block: the image of its input's initial marginal
        for t, ifs in self.factors:
            if t == "*" and ifs[0] == v and self.vars[ifs[2]].get("init") is not None:
                m, V = self.init_gauss(ifs[2])
                A = np.atleast_2d(self.const(ifs[1])).astype(float).reshape(d, self.dim[ifs[2]])
                return A @ m, A @ V @ A.T
        raise ValueError(f"variable {v} sits on a mean-field Gaussian node and has no Normal / MvNormal @initialization marginal")

    def _check_supported(self):
        for t, ifs in self.factors:
            if t in GAUSS_COV or t in GAUSS_PREC:
                scalar_data = self.kind[ifs[2]] in ("data", "gcvprec") and self.dim[ifs[0]] == 1
                if self.kind[ifs[2]] != "constant" and not scalar_data and not (t in GAUSS_PREC and ifs[2] in self.prec_prior):
                    raise ValueError(f"{t}: third interface must be a constant (or a Wishart / Gamma variable on a precision node; scalar nodes: a data variable)")
            elif t == "*":
                if self.kind[ifs[1]] != "constant":
                    raise ValueError("`*`: the matrix must be a constant")
            elif t == "+" or t in PRIORS or t == "GCVZ":
                pass
            elif t == "GCV":
                pass
            elif t == "NormalMixture":
                z = ifs[1]
                if z not in self.cat or self.kind[z] != "random":
                    raise ValueError("NormalMixture: the switch must be a random variable with a Categorical prior")
                K = (len(ifs) - 2) // 2
                for k in ifs[2:2 + K]:
                    if self.kind[k] != "random":
                        raise ValueError("NormalMixture: the means must be random variables")
                for k in ifs[2 + K:]:
                    if self.kind[k] == "random" and k not in self.prec_prior:
                        raise ValueError("NormalMixture: a random precision needs a Wishart / Gamma prior")
            elif t in ("Categorical", "Bernoulli"):
                if self.kind[ifs[1]] == "random" and ifs[1] not in self.dir:
                    raise ValueError("Categorical: a random probability vector needs a Dirichlet prior")
                if not self.mixtures:
                    raise ValueError(f"node {t} is not part of the Gaussian tree family")
            elif t in ("Dirichlet", "Beta"):
                if any(self.kind[x] != "constant" for x in ifs[1:]):
                    raise ValueError("Dirichlet / Beta: constant parameters expected")
                if not self.mixtures:
                    raise ValueError(f"node {t} is not part of the Gaussian tree family")
            else:
                raise ValueError(f"node {t} is not part of the Gaussian tree family")

    def alpha0(self, sv):
        """prior concentrations of a Dirichlet / Beta variable"""
        a = self.dir[sv]
        return np.array([float(np.ravel(self.const(x))[0]) for x in a]) if isinstance(a, tuple) else np.ravel(self.const(a)).astype(float)

    def const(self, v):
        x = np.asarray(self.vars[v]["value"], float)
        r, c = int(self.vars[v]["rows"]), int(self.vars[v]["cols"])
        return x.reshape(r, c) if c > 1 else x.reshape(r)

    def init_q(self, v):
        """the `@initialization` marginal of a precision variable as (nu, V) of a Wishart (a Gamma(a, b) is Wishart_1(2a, 1/(2b)))"""
        ini = self.vars[v].get("init")
        if ini is None:
            raise ValueError("a precision variable needs an @initialization marginal")
        p = np.asarray(ini["params"], float)
        d = self.dim[v]
        if ini["family"] == "gamma":
            return 2.0 * p[0], np.array([[1.0 / (2.0 * p[1])]])
        if ini["family"] == "wishart":
            return float(p[0]), p[1:].reshape(d, d)
        raise ValueError("unsupported initial marginal for a precision variable")

    def prior_q(self, v):
        t, ifs = self.factors[self.prec_prior[v]]
        a, b = self.const(ifs[1]), self.const(ifs[2])
        if t == "Wishart":
            return float(a.ravel()[0]), np.asarray(b, float).reshape(self.dim[v], self.dim[v])
        shape, par = float(a.ravel()[0]), float(np.ravel(b)[0])
        rate = par if t == "GammaShapeRate" else 1.0 / par
        return 2.0 * shape, np.array([[1.0 / (2.0 * rate)]])


def infer(dump, data, iterations=1, free_energy=True):
    """Returns dict(mean={var: m}, cov={var: V}, fe=[per iteration], q_prec={var: (nu, V)}, counters=dict(rule_calls, products, marginals))
    for ONE replica.  `data`: {variable id: vector}; a vector holding NaN is a `missing` observation: its node sends nothing and its Bethe terms cancel."""
    g = TreeGraph(dump)
    nv = len(g.vars)

    def value(v):   # clamped value of a data / constant variable, or of a deterministic function of such
        if g.kind[v] == "constant":
            return np.atleast_1d(g.const(v)).astype(float)
        if v in g.derived:
            t, ifs = g.factors[g.derived[v]]
            if t == "+":
                return value(ifs[1]) + value(ifs[2])
            A = np.atleast_2d(g.const(ifs[1])).astype(float).reshape(g.dim[ifs[0]], g.dim[ifs[2]])
            return A @ value(ifs[2])
        return np.atleast_1d(np.asarray(data[v], float))

    qW = {v: g.init_q(v) if g.vars[v].get("init") else g.prior_q(v) for v in g.prec_prior}
    # Mean field between the Gaussian interfaces of a node (q(out) q(μ)): the rule toward one interface reads the MARGINAL of the other —
    # MvNormalMeanCovariance(:out)(q_μ, q_Σ) = N(mean(q_μ), Σ), (:μ) alike — so the marginals of those variables are state, started from the @initialization
    # marginals.  Update order (ASSUMED — the reference's reactive order is not reproducible without it): every rule of an iteration reads the marginals
    # of the PREVIOUS iteration, then all marginals are replaced.  The fixed point does not depend on the order: tests/test_tree_oracle.py holds it to the
    # closed form of Gaussian mean field (exact means, blocks of the joint precision).
    qx = {}
    for fi, (t, ifs) in enumerate(g.factors):
        if g.mf[fi]:
            for v in ifs[:2]:
                qx[v] = g.init_gauss(v)
    # mixtures: the switch's rule reads the marginals of the means (and of a random `out`), q(p) and q(s) of the previous iteration
    for mx in g.mixtures:
        for v in list(mx["m"]) + [mx["out"]]:
            if g.gauss[v] and v not in qx:
                qx[v] = g.init_gauss(v)
    qs = {}
    for sv in g.dir:
        ini = g.vars[sv].get("init")
        qs[sv] = np.asarray(ini["params"] if ini is not None and ini["family"] == "dirichlet" else g.alpha0(sv), float).copy()

    # GCV nodes: the state of γ(z) = exp(−(κ z + ω)) under q(z): (E γ, E log γ), started from the `@initialization` marginal of z
    def gamma_of(gc, m, v):
        return math.exp(-gc["omega"] - gc["kappa"] * m + 0.5 * gc["kappa"] ** 2 * v), -(gc["kappa"] * m + gc["omega"])
    gstate = {}
    for gc in g.gcv:
        m0, V0 = g.init_gauss(gc["z"])
        gstate[gc["gamma"]] = gamma_of(gc, float(m0[0]), float(V0[0, 0]))
    gh_x, gh_w = np.polynomial.hermite.hermgauss(g.gh_points)
    gh_w = gh_w / math.sqrt(math.pi)

    def elog_s(sv, alphas):   # E log s of a Dirichlet variable, log p of a constant probability vector
        if g.kind[sv] == "constant":
            return np.log(np.ravel(g.const(sv)).astype(float))
        return digamma(alphas[sv]) - digamma(np.sum(alphas[sv]))
    fe_hist = []
    out = None
    for _ in range(max(1, int(iterations))):
        What = {v: qW[v][0] * qW[v][1] for v in qW}
        # q(z) of every mixture node from the marginals of the previous iteration
        pi = {}
        for mx in g.mixtures:
            o, z = mx["out"], mx["z"]
            d = g.dim[o]
            yo, Co = (qx[o][0], qx[o][1]) if g.gauss[o] else (value(o), np.zeros((d, d)))
            els = elog_s(g.cat[z], qs)
            lg = np.empty(len(mx["m"]))
            for k, (mk, pk) in enumerate(zip(mx["m"], mx["p"])):
                if pk in qW:
                    nu, V = qW[pk]
                    elw, Wk = mvdigamma(0.5 * nu, d) + d * math.log(2.0) + np.linalg.slogdet(V)[1], What[pk]
                else:
                    Wk = np.atleast_2d(g.const(pk)).astype(float).reshape(d, d)
                    elw = np.linalg.slogdet(Wk)[1]
                r = yo - qx[mk][0]
                lg[k] = els[k] - 0.5 * (d * LOG2PI - elw + np.trace(Wk @ (np.outer(r, r) + qx[mk][1] + Co)))
            w = np.exp(lg - np.max(lg))
            pi[z] = w / np.sum(w)
        # rule calls as the reference's trace counts them for a run WITHOUT the free energy: the messages the requested marginals pull in — the
        # marginals of the variables a user can name, i.e. not the anonymous output of a deterministic node (`B * x[t]`).  The remaining messages
        # (toward such outputs) are formed for the Bethe terms only and are not counted, as in oracle/rxoracle.c.
        counters = dict(rule_calls=0, products=0, marginals=0, on=True)
        det_outs = {ifs[0] for t, ifs in g.factors if t in ("*", "+")}
        f2v, v2f = {}, {}
        gcv_elq = {}

        def noise_of(fi):
            t, ifs = g.factors[fi]
            third = ifs[2]
            if third in g.prec_prior:
                W = What[third]
                return np.linalg.inv(W), W
            if g.kind[third] == "gcvprec":
                W = np.array([[gstate[third][0]]])
                return 1.0 / W, W
            M = np.atleast_2d(g.const(third) if g.kind[third] == "constant" else value(third)).astype(float)   # (a data-valued scalar variance: @autoupdates)
            if M.shape[0] != M.shape[1]:
                M = M.reshape(g.dim[ifs[0]], g.dim[ifs[0]])
            return (M, np.linalg.inv(M)) if t in GAUSS_COV else (np.linalg.inv(M), M)

        def msg_v2f(v, fi, k):
            key = (v, fi, k)
            if key not in v2f:
                ins = [msg_f2v(gf, gk) for gf, gk in g.nbrs[v] if (gf, gk) != (fi, k)]
                ins = [m for m in ins if m is not None]
                if not ins:
                    v2f[key] = None   # an improper (uniform) message: the variable has no other neighbour
                elif len(ins) == 1:
                    v2f[key] = ins[0]
                else:
                    xi, L = ins[0].wp()
                    for m in ins[1:]:
                        x2, L2 = m.wp()
                        xi, L = xi + x2, L + L2
                        counters["products"] += counters["on"]
                    v2f[key] = Msg("wp", xi, L)
            return v2f[key]

        def wp0(m, d):   # a missing (uniform) message carries no information
            return (np.zeros(d), np.zeros((d, d))) if m is None else m.wp()

        def additive(m, Sigma, W):
            if m.form == "mv":
                return Msg("mv", m.a, m.B + Sigma)
            G = np.linalg.inv(m.B + W)
            return Msg("wp", W @ G @ m.a, _sym(m.B @ G @ W))

        def msg_f2v(fi, k):
            key = (fi, k)
            if key in f2v:
                return f2v[key]
            t, ifs = g.factors[fi]
            res = None
            if t in GAUSS_COV or t in GAUSS_PREC:
                other = ifs[1 - k]
                Sigma, W = noise_of(fi)
                if fi in g.weight:   # a component of a mixture node: the Gaussian node's message with its log-potential weighted by q(z = k)
                    wk = pi[g.weight[fi][0]][g.weight[fi][1]]
                    src = qx[other][0] if g.gauss[other] else value(other)
                    res = Msg("wp", wk * (W @ src), wk * W)
                elif g.mf[fi]:
                    res = Msg("mv", qx[other][0], Sigma)
                elif g.gauss[other]:
                    m = msg_v2f(other, fi, 1 - k)
                    res = None if m is None else additive(m, Sigma, W)   # an unobserved leaf on the other side: nothing to pass on
                else:   # a `missing` observation (NaN) sends nothing
                    res = None if np.any(np.isnan(value(other))) else Msg("mv", value(other), Sigma)
            elif t == "GCVZ":
                gc = next(c for c in g.gcv if c["fz"] == fi)
                xy, Ly = wp0(msg_v2f(gc["y"], gc["fa"], 0), 1)
                xx, Lx = wp0(msg_v2f(gc["x"], gc["fa"], 1), 1)
                gam = gstate[gc["gamma"]][0]
                l11, l22 = float(Ly[0, 0]) + gam, float(Lx[0, 0]) + gam
                det = l11 * l22 - gam * gam
                v11, v22, v12 = l22 / det, l11 / det, gam / det
                m1, m2 = v11 * xy[0] + v12 * xx[0], v12 * xy[0] + v22 * xx[0]
                psi = (m1 - m2) ** 2 + v11 + v22 - 2.0 * v12
                a_, b_, c_ = gc["kappa"], psi * math.exp(-gc["omega"]), -gc["kappa"]
                gcv_elq[gc["z"]] = (a_, b_, c_, fi)
                epts = math.sqrt(2.0) * gh_x                                # what the neighbours of z see: the message's own Gaussian moments (no other message enters)
                ecs = gh_w * np.exp(-0.5 * (a_ * epts + b_ * np.exp(c_ * epts)) + 0.5 * epts * epts)
                em = float(np.sum(epts * ecs) / np.sum(ecs))
                ev = float(np.sum(ecs * (epts - em) ** 2) / np.sum(ecs))
                res = Msg("mv", [em], [[ev]])
            elif t == "*":
                A = np.atleast_2d(g.const(ifs[1])).astype(float)
                if A.shape != (g.dim[ifs[0]], g.dim[ifs[2]]):
                    A = A.reshape(g.dim[ifs[0]], g.dim[ifs[2]])
                src = msg_v2f(ifs[2], fi, 2) if k == 0 else msg_v2f(ifs[0], fi, 0)
                if src is None:
                    res = None
                elif k == 0:
                    m, V = src.mv()
                    res = Msg("mv", A @ m, _sym(A @ V @ A.T))
                else:
                    xi, L = src.wp()
                    res = Msg("wp", A.T @ xi, _sym(A.T @ L @ A))
            elif t == "+":
                o, a, b = ifs
                if k == 0:
                    srcs = [msg_v2f(x, fi, kk) if g.gauss[x] else Msg("mv", value(x), np.zeros((g.dim[x], g.dim[x]))) for kk, x in ((1, a), (2, b))]
                    if any(m is None for m in srcs):
                        res = None
                    else:
                        parts = [m.mv() for m in srcs]
                        res = Msg("mv", parts[0][0] + parts[1][0], parts[0][1] + parts[1][1])
                else:
                    oth, kk = (b, 2) if k == 1 else (a, 1)
                    mo = msg_v2f(o, fi, 0)
                    if mo is None or (g.gauss[oth] and msg_v2f(oth, fi, kk) is None):
                        res = None
                    elif g.gauss[oth] and mo.form == "mv":
                        m2, V2 = msg_v2f(oth, fi, kk).mv()
                        res = Msg("mv", mo.a - m2, mo.B + V2)
                    elif g.gauss[oth]:
                        # the message from `out` in precision form stays in it: N(m_out − m2, V_out + V2) = (ξ', Λ') with Λ' = Λo (Λo + W2)⁻¹ W2,
                        # ξ' = W2 (Λo + W2)⁻¹ (ξo + ξ2) − ξ2 — also where Λo is rank-deficient (an observation map with fewer rows than columns behind
                        # the `+`) and `mean_cov(m_out)`, which the reference's rule calls, does not exist
                        x2, W2 = msg_v2f(oth, fi, kk).wp()
                        G = np.linalg.inv(mo.B + W2)
                        res = Msg("wp", W2 @ G @ (mo.a + x2) - x2, _sym(mo.B @ G @ W2))
                    elif mo.form == "mv":
                        res = Msg("mv", mo.a - value(oth), mo.B)
                    else:
                        res = Msg("wp", mo.a - mo.B @ value(oth), mo.B)
            else:
                raise ValueError(t)
            if res is not None:
                counters["rule_calls"] += counters["on"]
            f2v[key] = res
            return res

        # ---- marginals ----
        mean, cov = {}, {}
        for v in [v for v in range(nv) if v not in det_outs] + [v for v in range(nv) if v in det_outs]:
            if not g.gauss[v]:
                continue
            counters["on"] = v not in det_outs or v in qx   # (a mean-field rule reads this marginal: it is computed whoever asks)
            ins = [m for m in (msg_f2v(fi, k) for fi, k in g.nbrs[v]) if m is not None]
            if not ins:
                raise ValueError(f"variable {v} receives no message")
            if len(ins) == 1:   # one message: the marginal is that message as it stands (no round trip of a covariance through its inverse)
                mean[v], cov[v] = (np.array(a, float) for a in ins[0].mv())
            else:
                xi, L = ins[0].wp()
                xi, L = xi.copy(), L.copy()
                for m in ins[1:]:
                    x2, L2 = m.wp()
                    xi, L = xi + x2, L + L2
                V = _sym(_inverse(L))
                mean[v], cov[v] = V @ xi, V
            if v in gcv_elq:   # the volatility input of a GCV node: the ELQ message times the product of ALL other messages, moment-matched by cubature against that product
                a_, b_, c_, fz = gcv_elq[v]
                fwd = msg_v2f(v, fz, 0)
                if fwd is None:
                    raise ValueError("GCV: z receives no Gaussian message to moment-match against")
                zm, fzv = (float(np.ravel(a)[0]) for a in fwd.mv())
                pts = zm + math.sqrt(2.0 * fzv) * gh_x
                cs = gh_w * np.exp(-0.5 * (a_ * pts + b_ * np.exp(c_ * pts)))
                qm = float(np.sum(pts * cs) / np.sum(cs))
                mean[v], cov[v] = np.array([qm]), np.array([[float(np.sum(cs * (pts - qm) ** 2) / np.sum(cs))]])
            counters["marginals"] += counters["on"] and v not in det_outs
        counters["on"] = False

        # ---- node-local joints of the Gaussian nodes with two random interfaces; residual second moments of every Gaussian node ----
        def node_moments(fi):
            """E[(out − μ)(out − μ)ᵀ] and the entropy of the node's Gaussian cluster under the node-local marginal"""
            t, ifs = g.factors[fi]
            o, mu = ifs[0], ifs[1]
            Sigma, W = noise_of(fi)
            d = g.dim[o]
            if g.mf[fi]:   # E[rrᵀ] under q(out) q(μ); the node's share of the entropies: both clusters
                r = mean[o] - mean[mu]
                return cov[o] + cov[mu] + np.outer(r, r), _entropy(cov[o]) + _entropy(cov[mu]), None
            if g.gauss[o] and g.gauss[mu]:
                xo, Lo = wp0(msg_v2f(o, fi, 0), d)
                xm, Lm = wp0(msg_v2f(mu, fi, 1), d)
                Lj = np.block([[Lo + W, -W], [-W, Lm + W]])
                Vj = _sym(np.linalg.inv(Lj))
                mj = Vj @ np.concatenate([xo, xm])
                D = np.hstack([np.eye(d), -np.eye(d)])
                r = D @ mj
                return D @ Vj @ D.T + np.outer(r, r), _entropy(Vj), (mj, Vj)
            if g.gauss[o] or g.gauss[mu]:
                rv, cv = (o, mu) if g.gauss[o] else (mu, o)
                r = mean[rv] - value(cv)
                if np.any(np.isnan(r)):   # `missing`: the clamped side becomes a predicted variable — U − H[q(y, x)] + 0·H[q(y)] = −H[q(x)]
                    return None, _entropy(cov[rv]), None
                return cov[rv] + np.outer(r, r), _entropy(cov[rv]), None
            r = value(o) - value(mu)
            if np.any(np.isnan(r)):
                return None, 0.0, None
            return np.outer(r, r), 0.0, None

        # ---- q(W) updates (mean field): Wishart(ν0 + n, (S0⁻¹ + Σ E[rrᵀ])⁻¹) ----
        stats = {v: [0, np.zeros((g.dim[v], g.dim[v]))] for v in qW}
        moments = {}
        for fi, (t, ifs) in enumerate(g.factors):
            if t in GAUSS_COV or t in GAUSS_PREC:
                moments[fi] = node_moments(fi)
                if ifs[2] in qW:
                    if moments[fi][0] is None:
                        raise ValueError("`missing` observations under a random precision are not part of the family")
                    wk = pi[g.weight[fi][0]][g.weight[fi][1]] if fi in g.weight else 1.0
                    stats[ifs[2]][0] += wk
                    stats[ifs[2]][1] += wk * moments[fi][0]
        gnew = {gc["gamma"]: gamma_of(gc, float(mean[gc["z"]][0]), float(cov[gc["z"]][0, 0])) for gc in g.gcv}
        qs_new = {sv: g.alpha0(sv).astype(float) + sum(pi[z] for z in pi if g.cat[z] == sv) for sv in qs}
        qnew = {}
        for v in qW:
            nu0, S0 = g.prior_q(v)
            Vi = np.linalg.inv(S0) + _sym(stats[v][1])
            qnew[v] = (nu0 + stats[v][0], _sym(np.linalg.inv(Vi)))

        # ---- Bethe free energy: q(x…) of this sweep, q(W) as just updated ----
        if free_energy:
            F = 0.0
            for fi, (t, ifs) in enumerate(g.factors):
                if t in GAUSS_COV or t in GAUSS_PREC:
                    d = g.dim[ifs[0]]
                    E, H, _ = moments[fi]
                    if E is None:
                        F += -H
                        continue
                    third = ifs[2]
                    if third in qW:
                        nu, V = qnew[third]
                        Elogdet = mvdigamma(0.5 * nu, d) + d * math.log(2.0) + np.linalg.slogdet(V)[1]
                        Wm = nu * V
                    elif g.kind[third] == "gcvprec":   # the GCV average energy with the NEW q(z)
                        Wm, Elogdet = np.array([[gnew[third][0]]]), gnew[third][1]
                    else:
                        _, Wm = noise_of(fi)
                        Elogdet = np.linalg.slogdet(Wm)[1]
                    wk = pi[g.weight[fi][0]][g.weight[fi][1]] if fi in g.weight else 1.0
                    F += wk * 0.5 * (d * LOG2PI - Elogdet + np.trace(Wm @ E)) - H
                elif t in ("*", "+") and fi in g.derived.values():
                    pass   # all interfaces clamped: the point entropies cancel (CountingReal bookkeeping)
                elif t == "*":
                    F += -_entropy(cov[ifs[2]])
                elif t == "+":
                    ins = [x for x in ifs[1:] if g.gauss[x]]
                    if len(ins) == 2:   # joint of the two inputs: q(in1, in2) ∝ m(in1) m(in2) m_out(in1 + in2)
                        dd = g.dim[ifs[0]]
                        x1, L1 = wp0(msg_v2f(ifs[1], fi, 1), dd)
                        x2, L2 = wp0(msg_v2f(ifs[2], fi, 2), dd)
                        xo, Lo = wp0(msg_v2f(ifs[0], fi, 0), dd)
                        Lj = np.block([[L1 + Lo, Lo], [Lo, L2 + Lo]])
                        F += -_entropy(_sym(np.linalg.inv(Lj)))
                    elif len(ins) == 1:
                        F += -_entropy(cov[ins[0]])
                elif t in PRIORS:
                    v = ifs[0]
                    d = g.dim[v]
                    nu0, S0 = g.prior_q(v)
                    nu, V = qnew[v]
                    ldV = np.linalg.slogdet(V)[1]
                    Elw = mvdigamma(0.5 * nu, d) + d * math.log(2.0) + ldV
                    # average energy of the Wishart prior node under q(W), and −H[q(W)]
                    U = -(0.5 * (nu0 - d - 1.0) * Elw - 0.5 * np.trace(np.linalg.inv(S0) @ (nu * V)) - 0.5 * nu0 * d * math.log(2.0) -
                          0.5 * nu0 * np.linalg.slogdet(S0)[1] - mvlgamma(0.5 * nu0, d))
                    Hq = 0.5 * (d + 1.0) * ldV + 0.5 * d * (d + 1.0) * math.log(2.0) + mvlgamma(0.5 * nu, d) - 0.5 * (nu - d - 1.0) * mvdigamma(0.5 * nu, d) + 0.5 * nu * d
                    # prior node U − H[q(W)]; each of the n likelihood nodes carries −H[q(W)] for its cluster (W) and the variable term is
                    # (degree − 1) H[q(W)] = n H[q(W)]: they cancel
                    F += U - Hq
            for v in range(nv):
                if g.gauss[v]:
                    F += (len(g.nbrs[v]) - 1) * _entropy(cov[v])
            for gc in g.gcv:   # the node's second cluster: −H[q(z)]
                F += -_entropy(cov[gc["z"]])
            # Categorical nodes: −Σ_k π_k E log s_k with the new q(s); −H[q(z)] (node terms −2H, variable term +H); Dirichlet prior node U − H[q(s)]
            for z, w in pi.items():
                F += -float(np.dot(w, elog_s(g.cat[z], qs_new))) + float(np.sum(w[w > 0.0] * np.log(w[w > 0.0])))
            for sv, al in qs_new.items():
                a0 = g.alpha0(sv).astype(float)
                els = digamma(al) - digamma(np.sum(al))
                logB = lambda a: float(np.sum(gammaln(a)) - gammaln(np.sum(a)))
                U = logB(a0) - float(np.dot(a0 - 1.0, els))
                Hs = logB(al) + (np.sum(al) - len(al)) * float(digamma(np.sum(al))) - float(np.dot(al - 1.0, digamma(al)))
                F += U - Hs
            fe_hist.append(float(F))
        qW = qnew
        gstate = gnew
        qs = qs_new
        qx = {v: (mean[v].copy(), cov[v].copy()) for v in qx}
        counters.pop("on")
        out = dict(mean=mean, cov=cov, joints={fi: m[2] for fi, m in moments.items() if m[2] is not None}, counters=counters)
    out["fe"] = fe_hist
    out["q_prec"] = qW
    out["q_dir"] = qs
    out["q_cat"] = pi
    return out
