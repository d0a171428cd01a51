"""TreeEngine — the level-scheduled node-array executor behind the C ABI (include/rxhip.h rxhip_tree_*): sum-product / mean-field VMP on ANY
acyclic Gaussian factor graph (what `rxhip_create` falls through to when no pattern-matched family fits), and `rule_eval`, the single-rule A/B
hook.  A graph comes from `rxhip.graph.GraphBuilder` (or a dump of the Julia plugin)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import RxHipError, c_double_p, c_int64_p


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class TreeEngine:
    """One compiled graph × n_replicas independent copies (different data, same constants)."""

    def __init__(self, gb, n_replicas=1, device=-1, stream=None, force_executor=True, allow_missing=False):
        """gb: GraphBuilder.  force_executor=False goes through rxhip_create (pattern matcher first; raises if a specialised engine took the graph).
        allow_missing: NaN in the data is a `missing` observation (rxhip_graph_desc.allow_missing): its node sends nothing, its Bethe terms cancel."""
        L = _lib.lib()
        g, keep = gb.tables(n_replicas=n_replicas, allow_missing=allow_missing)
        self._keep = (g, keep)
        self._h = ctypes.c_void_p()
        if force_executor:
            st = L.rxhip_tree_create(ctypes.byref(g), int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(self._h))
        else:
            st = L.rxhip_create(ctypes.byref(g), 0, int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(self._h))
        if st != _lib.OK:
            msg = L.rxhip_lowering_error().decode()
            if self._h:
                L.rxhip_destroy(self._h)
            self._h = None
            raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
        self.gb, self.n_replicas = gb, int(n_replicas)
        info = _lib.TreeInfo()
        if L.rxhip_tree_get_info(self._h, ctypes.byref(info)) != _lib.OK:
            L.rxhip_destroy(self._h)
            self._h = None
            raise RxHipError(_lib.ERR_BADARG, "the pattern matcher took this graph: not an engine of the node-array executor")
        self.info = {k: int(getattr(info, k)) for k, _ in _lib.TreeInfo._fields_ if k != "last_iteration_ms"}
        self._iters = 0

    def allreduce_free_energy(self, comm):
        """Sum the last run's per-iteration free energies over the ranks of `comm` (rxhip.Communicator), in rank order, in place on the device —
        replicas shard over GPUs with no other exchange (include/rxhip.h rxhip_allreduce_free_energy)."""
        self._chk(_lib.lib().rxhip_allreduce_free_energy(self._h, ctypes.c_void_p(int(getattr(comm, "handle", comm) or 0))))

    def last_iteration_ms(self):
        """device time of the last run ÷ its iterations"""
        info = _lib.TreeInfo()
        self._chk(_lib.lib().rxhip_tree_get_info(self._h, ctypes.byref(info)))
        return float(info.last_iteration_ms)

    def continue_runs(self, on=True):
        """later run() calls go on from the q(W) the previous run ended with instead of the `@initialization` marginals (rxhip_tree_continue): k calls of
        run(1) then equal one run(k) — how a driver that takes one VMP iteration per call (batch.jl:391-430) uses the engine"""
        self._chk(_lib.lib().rxhip_tree_continue(self._h, int(bool(on))))

    def _chk(self, st):
        if st != _lib.OK:
            raise RxHipError(st, _lib.lib().rxhip_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rxhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_data(self, variables, values):
        """variables: data variable ids; values: [replica][Σ rows] (the rows of the variables side by side, in list order)"""
        v = np.ascontiguousarray(variables, dtype=np.int64)
        x = _c(values).reshape(self.n_replicas, -1)
        rows = int(sum(self.gb.rows[i] for i in v))
        if x.shape[1] != rows:
            raise ValueError(f"values must be [replicas][{rows}]")
        self._chk(_lib.lib().rxhip_tree_set_data(self._h, v.ctypes.data_as(c_int64_p), len(v), x.ctypes.data_as(c_double_p)))

    def run(self, iterations=1, free_energy=True):
        self._chk(_lib.lib().rxhip_run(self._h, int(iterations), int(bool(free_energy))))
        self._iters = int(iterations)

    def marginals(self, variables):
        """{variable: (mean [replica][d], cov [replica][d][d])}"""
        v = np.ascontiguousarray(variables, dtype=np.int64)
        R = self.n_replicas
        dims = [self.gb.rows[i] for i in v]
        mean, cov = np.empty(R * sum(dims)), np.empty(R * sum(d * d for d in dims))
        self._chk(_lib.lib().rxhip_tree_get_marginals(self._h, v.ctypes.data_as(c_int64_p), len(v), mean.ctypes.data_as(c_double_p), cov.ctypes.data_as(c_double_p)))
        out, mo, co = {}, 0, 0
        for i, d in zip(v, dims):
            out[int(i)] = (mean[mo:mo + R * d].reshape(R, d), cov[co:co + R * d * d].reshape(R, d, d))
            mo += R * d
            co += R * d * d
        return out

    def precision(self, variable):
        """q(W) of a precision variable: (nu [replica], V [replica][d][d])"""
        d, R = self.gb.rows[variable], self.n_replicas
        nu, V = np.empty(R), np.empty((R, d, d))
        self._chk(_lib.lib().rxhip_tree_get_precision(self._h, int(variable), nu.ctypes.data_as(c_double_p), V.ctypes.data_as(c_double_p)))
        return nu, V

    def discrete(self, variable):
        """q(z = k) of a mixture node's switch, or the concentrations of q(s) = Dirichlet(α) of its probability vector: [replica][K]"""
        K = ctypes.c_int32(0)
        self._chk(_lib.lib().rxhip_tree_get_discrete(self._h, int(variable), None, ctypes.byref(K)))
        out = np.empty((self.n_replicas, K.value))
        self._chk(_lib.lib().rxhip_tree_get_discrete(self._h, int(variable), out.ctypes.data_as(c_double_p), None))
        return out

    def free_energy(self):
        """per iteration, summed over the replicas"""
        fe = np.empty(max(self._iters, 1))
        self._chk(_lib.lib().rxhip_get_free_energy(self._h, fe.ctypes.data_as(c_double_p)))
        return fe

    def free_energy_per_replica(self):
        fe = np.empty(self.n_replicas)
        self._chk(_lib.lib().rxhip_get_free_energy_per_chain(self._h, fe.ctypes.data_as(c_double_p)))
        return fe

    def counters(self):
        r, p, m = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(_lib.lib().rxhip_counters(self._h, ctypes.byref(r), ctypes.byref(p), ctypes.byref(m)))
        return dict(rule_calls=r.value, products=p.value, marginals=m.value)


def plan(gb, n_replicas=1, allow_missing=False):
    """The graph compiler alone (rxhip_tree_plan; host only, no GPU needed): dict of the schedule's static figures plus the reference-equivalent counts of one
    replica and iteration (rule_calls, products, marginals).  Raises RxHipError exactly where TreeEngine(gb) would refuse the graph."""
    g, keep = gb.tables(n_replicas=n_replicas, allow_missing=allow_missing)
    info = _lib.TreeInfo()
    rc, pr, mg = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    L = _lib.lib()
    st = L.rxhip_tree_plan(ctypes.byref(g), ctypes.byref(info), ctypes.byref(rc), ctypes.byref(pr), ctypes.byref(mg))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode() or L.rxhip_status_string(st).decode())
    out = {k: int(getattr(info, k)) for k, _ in _lib.TreeInfo._fields_ if k != "last_iteration_ms"}
    out.update(rule_calls=rc.value, products=pr.value, marginals=mg.value)
    return out


def rule_eval(node_type, iface, constant, msg, msg2=None, in_form="mv", out_form="mv", d_out=None, device=-1):
    """One message rule on the device for a batch (rxhip_rule_eval): msg = (a [n][d], B [n][d][d]) in `in_form` ('mv': mean / covariance, 'wp':
    weighted mean / precision).  Returns (a', B') in `out_form`."""
    a, B = _c(msg[0]), _c(msg[1])
    n, din_msg = a.shape
    c = _lib.RuleCall()
    c.node_type, c.iface, c.n = int(node_type), int(iface), n
    keep = [a, B]
    if node_type == _lib.NODE_MULTIPLY:
        A = _c(np.atleast_2d(constant))
        c.d_out, c.d_in = A.shape
        c.constant = A.ctypes.data_as(c_double_p)
        keep.append(A)
        dres = c.d_out if iface == 0 else c.d_in
    else:
        c.d_out = c.d_in = din_msg if d_out is None else d_out
        if constant is not None:
            C = _c(np.atleast_2d(constant))
            c.constant = C.ctypes.data_as(c_double_p)
            keep.append(C)
        dres = c.d_out
    c.in_form = 0 if in_form == "mv" else 1
    c.out_form = 0 if out_form == "mv" else 1
    c.in_a, c.in_B = a.ctypes.data_as(c_double_p), B.ctypes.data_as(c_double_p)
    if msg2 is not None:
        a2, B2 = _c(msg2[0]), _c(msg2[1])
        keep += [a2, B2]
        c.in2_a, c.in2_B = a2.ctypes.data_as(c_double_p), B2.ctypes.data_as(c_double_p)
    oa, oB = np.empty((n, dres)), np.empty((n, dres, dres))
    c.out_a, c.out_B = oa.ctypes.data_as(c_double_p), oB.ctypes.data_as(c_double_p)
    L = _lib.lib()
    st = L.rxhip_rule_eval(ctypes.byref(c), int(device))
    if st != _lib.OK:
        raise RxHipError(st, L.rxhip_lowering_error().decode() or L.rxhip_status_string(st).decode())
    return oa, oB
