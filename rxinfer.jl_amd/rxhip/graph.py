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


class GraphBuilder:
    def __init__(self):
        self.kind, self.rows, self.cols, self.coff = [], [], [], []
        self.ftype, self.fiface = [], []
        self.pool = []
        self._n = 0

    def _var(self, kind, rows, cols=1, coff=-1):
        self.kind.append(kind); self.rows.append(rows); self.cols.append(cols); self.coff.append(coff)
        return len(self.kind) - 1

    def randomvar(self, dim):
        return self._var(_lib.VARKIND_RANDOM, dim)

    def datavar(self, dim):
        return self._var(_lib.VARKIND_DATA, dim)

    def constvar(self, value):
        v = np.atleast_1d(np.asarray(value, dtype=np.float64))
        rows, cols = (v.shape[0], 1) if v.ndim == 1 else v.shape
        off = self._n
        self.pool.append(v.ravel())
        self._n += v.size
        return self._var(_lib.VARKIND_CONST, rows, cols, off)

    def mvnormal_mean_cov(self, out, mu, sigma):
        """out ~ MvNormal(μ = mu, Σ = sigma)  ->  MvNormalMeanCovariance (src/model/graphppl.jl:372-376)"""
        self.ftype.append(_lib.NODE_MVNORMAL_MEAN_COV); self.fiface.append((out, mu, sigma))

    def multiply(self, out, A, x):
        """out := A * x  ->  typeof(*) node with an anonymous output variable"""
        self.ftype.append(_lib.NODE_MULTIPLY); self.fiface.append((out, A, x))

    def tables(self, n_replicas=1, permute=None):
        ft = np.asarray(self.ftype, dtype=np.int32)
        fi = np.asarray(self.fiface, dtype=np.int64).reshape(-1, 3)
        if permute is not None:  # node order must not matter to the lowering
            ft, fi = ft[permute], fi[permute]
        arrs = dict(kind=np.asarray(self.kind, dtype=np.int32), rows=np.asarray(self.rows, dtype=np.int32),
                    cols=np.asarray(self.cols, dtype=np.int32), coff=np.asarray(self.coff, dtype=np.int64), ft=np.ascontiguousarray(ft),
                    fi=np.ascontiguousarray(fi), pool=np.concatenate(self.pool) if self.pool else np.zeros(1))
        g = _lib.GraphDesc()
        g.n_variables = len(self.kind)
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
        return g, arrs  # keep `arrs` alive while `g` is in use


def lgssm_graph(T, A, B, P, Q, m0, V0, prior_through_transition=False, A_of_t=None):
    """The graph GraphPPL builds for the benchmark notebook's model (cell 4) / mlgssm_test.jl:9-17."""
    A, B = np.asarray(A, float), np.asarray(B, float)
    d, dy = A.shape[0], B.shape[0]
    gb = GraphBuilder()
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(m0), gb.constvar(V0))
    xs, ys = [], []
    for t in range(T):
        if t > 0 or prior_through_transition:
            a = gb.randomvar(d)
            gb.multiply(a, gb.constvar(A if A_of_t is None else A_of_t(t)), x)
            xn = gb.randomvar(d)
            gb.mvnormal_mean_cov(xn, a, gb.constvar(P))
            x = xn
        b = gb.randomvar(dy)
        gb.multiply(b, gb.constvar(B), x)
        y = gb.datavar(dy)
        gb.mvnormal_mean_cov(y, b, gb.constvar(Q))
        xs.append(x); ys.append(y)
    return gb, xs, ys


def lower_lgssm(g):
    """Host-only lowering (no GPU): returns dict(d, dy, T, prior_through_transition, A, B, P, Q, m0, V0, state_var, data_var)."""
    L = _lib.lib()
    out = _lib.LgssmLowered()
    st = L.rxhip_graph_lower_lgssm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    d, dy, T = out.d, out.dy, out.T
    bufs = dict(A=np.empty((d, d)), B=np.empty((dy, d)), P=np.empty((d, d)), Q=np.empty((dy, dy)), m0=np.empty(d), V0=np.empty((d, d)))
    sv, dv = np.empty(T, dtype=np.int64), np.empty(T, dtype=np.int64)
    for k, v in bufs.items():
        setattr(out, k, v.ctypes.data_as(_lib.c_double_p))
    out.state_var = sv.ctypes.data_as(_lib.c_int64_p)
    out.data_var = dv.ctypes.data_as(_lib.c_int64_p)
    st = L.rxhip_graph_lower_lgssm(ctypes.byref(g), ctypes.byref(out))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode())
    return dict(d=d, dy=dy, T=T, prior_through_transition=bool(out.prior_through_transition), state_var=sv, data_var=dv, **bufs)


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
    low = lower_lgssm(g)
    eng = LGSSMEngine.__new__(LGSSMEngine)
    eng._h, eng.d, eng.dy, eng.T, eng.n_chains, eng.n_models = h, low["d"], low["dy"], low["T"], int(g.n_replicas or 1), 1
    eng._keep, eng._data_ref = [], None
    return eng
