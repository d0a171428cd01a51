#!/usr/bin/env python3
"""Writes tests/golden/graph_dumps/*.json.gz: the factor graphs GraphPPL builds for the reference's test models, in the
exchange format the Julia plugin's `dump_graph` emits (rxinfer.jl_amd/julia/HIPInferencePlugin.jl, "rxhip-graph-1") and
`rxhip.graph.GraphBuilder.from_dump` reads — including each node's factorisation clusters ("clusters": one id per interface, the
node's VariationalConstraintsFactorizationIndicesKey as the model's @constraints materialise it).  No Julia exists in the build image, so the dumps are written BY HAND from the
`@model` bodies, statement by statement (variables in creation order, one constant variable per use of a constant, one
anonymous random variable per `A * x` call — docs/src/manuals/model-specification.md:70,217-240):

  mlgssm.json.gz           test/models/statespace/mlgssm_test.jl:9-17,72-97    multivariate_lgssm_model, n = 1000
  ulgssm.json.gz           test/models/statespace/ulgssm_tests.jl:8-15,27-33   univariate_lgssm_model, n = 500
  gmm_univariate.json.gz   test/models/mixtures/gmm_univariate_tests.jl:7-26   Beta / Bernoulli spelling, n = 150
  gmm_multivariate.json.gz test/models/mixtures/gmm_multivariate_tests.jl:6-32,66-70   K = 3, d = 2, n = 500 (priors from the fixture)
  hgf_step.json.gz         test/models/statespace/hgf_tests.jl:9-41,51-54     one-step graph + @initialization + GCV meta

Run from the repo root:  python tests/golden/make_graph_dumps.py"""
import gzip
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "rxinfer.jl_amd"))
from rxhip import _lib  # noqa: E402
from rxhip.graph import GraphBuilder  # noqa: E402

OUT = os.path.join(HERE, "graph_dumps")


def write(name, gb, **kw):
    os.makedirs(OUT, exist_ok=True)
    with gzip.GzipFile(os.path.join(OUT, name + ".json.gz"), "wb", mtime=0) as f:  # mtime = 0: byte-reproducible
        f.write(json.dumps(gb.to_dump(**kw), ensure_ascii=False, separators=(",", ":")).encode())


def mlgssm():
    g = np.load(os.path.join(HERE, "mlgssm_stablerng1234.npz"))
    A, B, Q, P = g["A"], g["B"], g["state_noise"], g["obs_noise"]  # the test's names: Q state noise, P observation noise
    gb = GraphBuilder()
    x_prev = gb.randomvar(2, "x_prior")
    gb.node(_lib.NODE_MVNORMAL_MEAN_COV, x_prev, gb.constvar(g["prior_mean"], "constvar"), gb.constvar(g["prior_cov"], "constvar"))
    for i in range(g["y"].shape[0]):
        a = gb.randomvar(2, "anonymous")
        gb.node(_lib.NODE_MULTIPLY, a, gb.constvar(A, "constvar"), x_prev)
        x = gb.randomvar(2, f"x[{i + 1}]")
        gb.node(_lib.NODE_MVNORMAL_MEAN_COV, x, a, gb.constvar(Q, "constvar"))
        b = gb.randomvar(2, "anonymous")
        gb.node(_lib.NODE_MULTIPLY, b, gb.constvar(B, "constvar"), x)
        y = gb.datavar(2, f"y[{i + 1}]")
        gb.node(_lib.NODE_MVNORMAL_MEAN_COV, y, b, gb.constvar(P, "constvar"))
        x_prev = x
    write("mlgssm", gb.bethe())   # no @constraints: GraphPPL's default — the random interfaces of a node share one factor of q


def ulgssm():
    g = np.load(os.path.join(HERE, "ulgssm_stablerng123.npz"))
    gb = GraphBuilder()
    x_prev = gb.randomvar(1, "x_prior")
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x_prev, gb.constvar(float(g["prior_mean"]), "constvar"), gb.constvar(float(g["prior_var"]), "constvar"))
    for i in range(g["y"].size):
        x = gb.randomvar(1, f"x[{i + 1}]")
        gb.node(_lib.NODE_ADD, x, x_prev, gb.constvar(float(g["c"]), "constvar"))
        y = gb.datavar(1, f"y[{i + 1}]")
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, x, gb.constvar(float(g["obs_var"]), "constvar"))
        x_prev = x
    write("ulgssm", gb.bethe())


def gmm_univariate():
    n = 150
    gb = GraphBuilder()
    s = gb.randomvar(1, "s")
    gb.node(_lib.NODE_BETA, s, gb.constvar(1.0, "constvar"), gb.constvar(1.0, "constvar"))
    m, p = [], []
    for k, mean in enumerate((-2.0, 2.0)):
        mk = gb.randomvar(1, f"m[{k + 1}]")
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, mk, gb.constvar(mean, "constvar"), gb.constvar(1e3, "constvar"))
        pk = gb.randomvar(1, f"p[{k + 1}]")
        gb.node(_lib.NODE_GAMMA_SHAPE_RATE, pk, gb.constvar(0.01, "constvar"), gb.constvar(0.01, "constvar"))
        m.append(mk); p.append(pk)
    for i in range(n):
        z = gb.randomvar(1, f"z[{i + 1}]")
        gb.node(_lib.NODE_BERNOULLI, z, s)
        y = gb.datavar(1, f"y[{i + 1}]")
        gb.node(_lib.NODE_NORMAL_MIXTURE, y, z, *m, *p)
    # @initialization: q(s) = vague(Beta); q(m) = NormalMeanVariance(∓2, 1e3); q(p) = vague(GammaShapeRate) = (1, tiny)
    gb.initialize(s, _lib.INIT_DIRICHLET, (1.0, 1.0))
    for k, mean in enumerate((-2.0, 2.0)):
        gb.initialize(m[k], _lib.INIT_NORMAL, (mean, 1e3))
        gb.initialize(p[k], _lib.INIT_GAMMA, (1.0, 1e-12))
    write("gmm_univariate", gb.mean_field())   # gmm_univariate_tests.jl:63-72: MeanField() (and its spelled-out twin)


def gmm_multivariate():
    g = np.load(os.path.join(HERE, "mvgmm_stablerng43.npz"))
    K, d, n = 3, 2, g["y"].shape[0]
    gb = GraphBuilder()
    m, w = [], []
    for k in range(K):   # gmm_multivariate_tests.jl:11-24: m[k] then w[k] inside one loop
        mk = gb.randomvar(d, f"m[{k + 1}]")
        gb.node(_lib.NODE_MVNORMAL_MEAN_COV, mk, gb.constvar(g["prior_mean"][k], "constvar"), gb.constvar(g["prior_cov"], "constvar"))
        wk = gb.randomvar(d, f"w[{k + 1}]")
        gb.node(_lib.NODE_WISHART, wk, gb.constvar(float(g["wishart_nu"]), "constvar"), gb.constvar(g["wishart_scale"], "constvar"))
        m.append(mk); w.append(wk)
    s = gb.randomvar(K, "s")
    gb.node(_lib.NODE_DIRICHLET, s, gb.constvar(np.ones(K), "constvar"))
    for i in range(n):
        z = gb.randomvar(1, f"z[{i + 1}]")
        gb.node(_lib.NODE_CATEGORICAL, z, s)
        y = gb.datavar(d, f"y[{i + 1}]")
        gb.node(_lib.NODE_NORMAL_MIXTURE, y, z, *m, *w)
    for k in range(K):   # :66-70  q(s) = vague(Dirichlet, 3); q(m) = MvNormalMeanCovariance(b, 1e6·I); q(w) = Wishart(3, 1e2·I)
        gb.initialize(m[k], _lib.INIT_MVNORMAL, np.concatenate([g["init_mean"][k], np.ravel(g["prior_cov"])]))
        gb.initialize(w[k], _lib.INIT_WISHART, np.concatenate([[float(g["wishart_nu"])], np.ravel(g["wishart_scale"])]))
    gb.initialize(s, _lib.INIT_DIRICHLET, np.ones(K))
    write("gmm_multivariate", gb.mean_field())   # gmm_multivariate_tests.jl:72-76


def hgf_step():
    g = np.load(os.path.join(HERE, "hgf_stablerng42.npz"))
    gb = GraphBuilder()   # hgf_tests.jl:9-31, statement order
    zt_min = gb.randomvar(1, "zt_min")
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, zt_min, gb.datavar(1, "z_prev_mean"), gb.datavar(1, "z_prev_var"))
    xt_min = gb.randomvar(1, "xt_min")
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, xt_min, gb.datavar(1, "x_prev_mean"), gb.datavar(1, "x_prev_var"))
    zt = gb.randomvar(1, "zt")
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, zt, zt_min, gb.constvar(float(g["z_variance"]), "constvar"))
    xt = gb.randomvar(1, "xt")
    gb.node(_lib.NODE_GCV, xt, xt_min, zt, gb.constvar(float(g["kappa"]), "constvar"), gb.constvar(float(g["omega"]), "constvar"))
    y = gb.datavar(1, "y")
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, xt, gb.constvar(float(g["y_variance"]), "constvar"))
    gb.initialize(zt, _lib.INIT_NORMAL, (0.0, 5.0))   # :51-54
    gb.initialize(xt, _lib.INIT_NORMAL, (0.0, 5.0))
    gb.gh_points = 31                                  # GCVMetadata(GaussHermiteCubature(31)), :37-40
    gb.bethe()
    gcv = gb.ftype.index(_lib.NODE_GCV)
    gb.set_clusters(gcv, (0, 0, 1, 2, 3))              # :33-35  q(xt, zt, xt_min) = q(xt, xt_min)q(zt)
    write("hgf_step", gb, n_observations=int(g["y"].size))


if __name__ == "__main__":
    mlgssm(); ulgssm(); gmm_univariate(); gmm_multivariate(); hgf_step()
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})
